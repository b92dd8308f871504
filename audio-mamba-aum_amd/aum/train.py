"""Training / evaluation launcher with the reference's command line (/root/reference/src/run.py = "RUN":36-132 and
/root/reference/src/traintest.py = "TT"), re-hosted for one-process-per-GPU `torch.distributed` over RCCL:

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m aum.train \
        --model aum --model_type base --aum_type Fo-Bi --dataset audioset --data-train train.json \
        --data-val eval.json --label-csv class_labels_indices.csv --n_class 527 --lr 1e-4 -b 64 --n-epochs 5 \
        --freqm 48 --timem 192 --mixup 0.5 --loss BCE --metrics mAP --warmup True --exp-dir exp/base

What is the same: the flags and their defaults, Adam with the batch-scaled betas/eps (TT:25-33), the 1000-step
linear warm-up in 50-step stairs (TT:119-123), MultiStepLR (TT:72), BCE/CE, nan_to_num and skip-on-inf (TT:153-164),
per-epoch validation with mAP/AUC/d' (TT:188-218), and every file an experiment directory holds afterwards
(args.pkl, result.csv, progress.pkl, stats_*.pickle, predictions/*.csv, models/{best,latest}_*.pth).

What is MI355X-first: DataLoader workers only decode waveforms; mel + SpecAug + normalisation + noise run on the
GPU; bf16 autocast (no GradScaler needed) around the HIP mixer; DDP with a static graph and gradient-as-bucket-view.
Out of scope here (rejected with an error): --model ast, epic_sounds, flexible patch training, ImageNet init.
"""
import argparse
import ast as _ast
import datetime
import os
import pickle
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

_lit = _ast.literal_eval
EXP_SEED = 3949                                                           # RUN:28-30


def build_parser():
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    a = p.add_argument
    a("--exp-dir", type=str, default="")
    a("--exp-name", type=str, default="")
    a("-w", "--num-workers", default=4, type=int)
    a("--n-print-steps", type=int, default=100)
    a("--run_type", type=str, default="train", choices=["train", "eval"])
    a("--data-train", type=str, default="")
    a("--data-val", type=str, default="")
    a("--data-eval", type=str, default="")
    a("--label-csv", type=str, default="")
    a("--n_class", type=int, default=527)
    a("--dataset", type=str, default="audioset")
    a("--freqm", type=int, default=0)
    a("--timem", type=int, default=0)
    a("--mixup", type=float, default=0)
    a("--dataset_mean", type=float, default=-4.2677393)
    a("--dataset_std", type=float, default=4.5689974)
    a("--audio_length", type=int, default=1024)
    a("--noise", type=_lit, default="False")
    a("--melbins", type=int, default=128)
    a("--fshift", type=int, default=10)
    a("--sample_rate", type=int, default=16000, help="(new) all clips must have this rate; the mel tables are built for it")
    a("--model", type=str, default="aum")
    a("--model_type", type=str, default="base")
    a("--fpatch_size", type=int, default=16)
    a("--tpatch_size", type=int, default=16)
    a("--fstride", type=int, default=16)
    a("--tstride", type=int, default=16)
    a("--imagenet_pretrain", type=_lit, default="False")
    a("--aum_pretrain", type=_lit, default="False")
    a("--aum_pretrain_path", type=str, default=None)
    a("--aum_pretrain_fstride", type=int, default=16)
    a("--aum_pretrain_tstride", type=int, default=16)
    a("--if_continue_inf", type=_lit, default="True")
    a("--if_nan2num", type=_lit, default="True")
    a("--aum_drop_path", type=float, default=0)
    a("--if_cls_token", type=_lit, default="True")
    a("--use_middle_cls_token", type=_lit, default="True")
    a("--use_double_cls_token", type=_lit, default="False")
    a("--use_end_cls_token", type=_lit, default="False")
    a("--transpose_token_sequence", type=_lit, default="False")
    a("--if_random_cls_token_position", type=_lit, default="False")
    a("--if_random_token_rank", type=_lit, default="False")
    a("--aum_type", type=str, default="Fo-Bi")
    a("--lr", "--learning-rate", default=0.001, type=float)
    a("--optim", type=str, default="adam", choices=["sgd", "adam"])
    a("-b", "--batch-size", default=12, type=int)
    a("--n-epochs", type=int, default=1)
    a("--save_model", type=_lit, default="True")
    a("--bal", type=str, default=None)
    a("--metrics", type=str, default="mAP", choices=["acc", "mAP"])
    a("--loss", type=str, default="BCE", choices=["BCE", "CE"])
    a("--warmup", type=_lit, default="False")
    a("--lrscheduler_start", type=int, default=2)
    a("--lrscheduler_step", type=int, default=1)
    a("--lrscheduler_decay", type=float, default=0.5)
    a("--bs_scale_factor", type=int, default=1)
    a("--weight_decay", type=float, default=5e-7)
    a("--optim_path", type=str, default=None)
    a("--flexible_training", type=_lit, default="False")
    a("--mixed_precision", type=str, default="bf16", choices=["no", "bf16", "fp16"],
      help="(new) what `accelerate launch --mixed_precision` selects for the reference")
    a("--depth", type=int, default=24, help="(new) number of blocks; every published AuM size uses 24 (RUN:227-237)")
    a("--grad_compress", type=str, default="no", choices=["no", "bf16", "fp16"],
      help="(new) exchange the gradient buckets in 16 bits (DDP communication hook): 368 -> 184 MB per step for AuM-Base")
    a("--max-steps", type=int, default=0, help="(new) stop each epoch after this many steps (smoke runs)")
    return p


def check_scope(args):
    if args.model != "aum":
        raise NotImplementedError("--model ast (the transformer baseline) is outside the accelerated path")
    if args.dataset == "epic_sounds" or args.flexible_training or args.imagenet_pretrain or args.aum_drop_path:
        raise NotImplementedError("epic_sounds / flexible training / ImageNet init / drop-path are out of scope")
    if not args.if_cls_token or args.use_double_cls_token or args.if_random_cls_token_position or args.if_random_token_rank:
        raise NotImplementedError("no / double / randomly placed cls tokens are off the accelerated path")



def free_port():
    """a TCP port the kernel hands out as free right now (rendezvous of a world-size-1 process group)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


class Dist:
    """the few things the reference uses `accelerator` for"""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.cuda = torch.cuda.is_available()
        # device index modulo + AUM_DIST_BACKEND=gloo: dry run of the multi-rank path on a box with fewer GPUs than ranks
        self.dev_index = self.local_rank % torch.cuda.device_count() if self.cuda else 0
        self.device = torch.device("cuda", self.dev_index) if self.cuda else torch.device("cpu")
        if self.cuda:
            torch.cuda.set_device(self.device)
        # AUM_FORCE_DDP=1: the data-parallel machinery (process group, DistributedDataParallel reducer, bucket views, the gradient
        # exchange hook, the MIN-reduced finite flag) also at world size 1 -- how the RCCL path is exercised on a one-GPU box
        self.ddp = self.world > 1 or os.environ.get("AUM_FORCE_DDP", "0") == "1"
        if self.ddp and not dist.is_initialized():
            if self.world == 1:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", str(free_port()))      # two forced-DDP runs on one box must not meet on a fixed port
                os.environ.setdefault("RANK", "0")
                os.environ.setdefault("WORLD_SIZE", "1")
            backend = os.environ.get("AUM_DIST_BACKEND", "nccl" if self.cuda else "gloo")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=self.device)
            else:
                dist.init_process_group(backend)
        self.main = self.rank == 0

    def print(self, *a):
        if self.main:
            print(*a, flush=True)

    def gather(self, t):
        if self.world == 1:
            return t if t.dim() else t[None]
        t = t.contiguous() if t.dim() else t[None].contiguous()
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return torch.cat(out)

    def barrier(self):
        if self.world > 1:
            dist.barrier()


def build_model(args):
    from .model import AudioMamba, AUM_SIZES
    size = next((s for s in AUM_SIZES if s in args.model_type), None)
    if size is None:
        raise ValueError("unknown model type, model type should be one of [base, small, tiny] for aum")
    bimamba = {"Fo-Fo": "none", "Fo-Bi": "v1", "Bi-Bi": "v2"}.get(args.aum_type)
    if bimamba is None:
        raise ValueError("unknown aum type, aum type should be one of [Fo-Fo, Fo-Bi, Bi-Bi] for aum")
    model = AudioMamba(spectrogram_size=(args.melbins, args.audio_length), patch_size=(args.fpatch_size, args.tpatch_size),
                       strides=(args.fstride, args.tstride), depth=args.depth, embed_dim=AUM_SIZES[size], num_classes=args.n_class,
                       bimamba_type=bimamba, use_middle_cls_token=args.use_middle_cls_token,
                       use_end_cls_token=args.use_end_cls_token, transpose_token_sequence=args.transpose_token_sequence)
    if args.aum_pretrain:
        from .checkpoint import load_aum_checkpoint
        print(load_aum_checkpoint(model, args.aum_pretrain_path, args.aum_pretrain_fstride, args.aum_pretrain_tstride))
    return model


def compress_gradients(ddp, kind):
    """The gradient exchange of a DistributedDataParallel model (TT:39, TT:168).  Returns an ssi.GradHomes: call its after_backward() after
    every backward pass and the kernels write the parameter gradients straight into the reducer's buckets (no per-parameter copies).
    kind "bf16" / "fp16": each fp32 bucket is cast,
    all-reduced (RCCL ring over xGMI: per-link bound, so half the bytes is close to half the exchange time) and cast back into the
    bucket view; master weights and Adam stay fp32.  "no": the reducer's own fp32 all-reduce + mean.  A model with Bi-Bi (v2) blocks gets
    ssi.ddp_join_streams_hook around whichever exchange it is: its two backward streams are joined before the collective is ordered behind
    one of them (and only then do the blocks use their second stream under a process group)."""
    from mamba_ssm.ops import selective_scan_interface as ssi
    from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
    inner = {"no": default_hooks.allreduce_hook, "bf16": default_hooks.bf16_compress_hook, "fp16": default_hooks.fp16_compress_hook}[kind]
    if any(getattr(m, "bimamba_type", None) == "v2" for m in ddp.modules()):
        ssi.register_ddp_join_streams(ddp, None, inner)      # Bi-Bi blocks: two backward streams to join; marks THIS wrapper's blocks
    else:
        # "no" is registered too (the reducer's own exchange, as a hook): without a hook the reducer divides every gradient that already
        # lives in its bucket (ssi.adopt_grad_homes) by the world size one launch per parameter; the hook divides the bucket once
        if kind == "no":
            # (the reducer's C++ statement of the same hook: divide the bucket once, all-reduce -- no Python on the autograd thread; every
            # Python hook call holds back the launches behind it by ~30 us, 14 buckets per step)
            import torch.distributed as dist_
            ddp._register_builtin_comm_hook(dist_.BuiltinCommHookType.ALLREDUCE)
        else:
            ddp.register_comm_hook(None, inner)
    return ssi.GradHomes(ddp.module)


class Frontend:
    """waveform batch on the device -> (model input, frontend= argument of AudioMamba.forward): the mean-removed waveform plus
    this batch's augmentation draw (waveform -> tokens runs inside the model, one launch), or -- fused=False -- the augmented,
    normalised (B, T, F) log-mel and None."""

    def __init__(self, args, device, train):
        from .frontend import FbankTables, pad_fill
        self.args, self.train = args, train
        self.tables = FbankTables(device, sample_rate=args.sample_rate, num_mel_bins=args.melbins,
                                  frame_shift_ms=float(args.fshift))
        self.fill = pad_fill(args.dataset_mean, args.dataset_std)

    def __call__(self, wave, n_valid, fused=True):
        from .frontend import WaveInput, prepare_wave
        from .augment import draw_augmentation
        a = self.args
        aug = nz = None
        if self.train and (a.freqm or a.timem or a.noise):       # DL:206-228, applied in the log-mel kernel's own store
            aug, nz = draw_augmentation(wave.shape[0], a.audio_length, a.melbins, a.freqm, a.timem, a.noise, wave.device)
        wave, aug = prepare_wave(wave, n_valid, self.tables, aug)
        fe = WaveInput(self.tables, a.audio_length, a.dataset_mean, a.dataset_std, aug, nz)
        return (wave, fe) if fused else (fe.spectrogram(wave), None)


def make_loader(args, path, train, D):
    from .data import WaveformDataset
    win = int(args.sample_rate * 0.025)
    shift = int(args.sample_rate * args.fshift * 0.001)
    max_samples = win + (args.audio_length - 1) * shift              # exactly audio_length frames
    ds = WaveformDataset(path, args.label_csv, max_samples, mixup=args.mixup if train else 0.0, sample_rate=args.sample_rate)
    bs = args.batch_size if train else args.batch_size * 2                 # RUN:190
    sampler = None
    if train and args.bal == "bal":                                        # RUN:173-181
        w = np.loadtxt(path[:-5] + "_weight.csv", delimiter=",")
        g = torch.Generator().manual_seed(EXP_SEED + D.rank)
        sampler = torch.utils.data.WeightedRandomSampler(w, len(w) // D.world, replacement=True, generator=g)
    elif D.world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(ds, D.world, D.rank, shuffle=train, drop_last=False)
    g = torch.Generator().manual_seed(EXP_SEED + 1000 * D.rank + (1 if train else 0))    # worker base seeds differ per rank
    return torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=(train and sampler is None), sampler=sampler,
                                       num_workers=args.num_workers, pin_memory=D.cuda, drop_last=False,
                                       generator=g, worker_init_fn=_seed_worker)


def _seed_worker(worker_id):
    """python / numpy generators of a loader worker follow torch's per-worker seed (mix-up partners and lambdas, DL:104-129)"""
    seed = torch.initial_seed() % (2 ** 32)
    random.seed(seed)
    np.random.seed(seed)


def _autocast(args, D):
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}.get(args.mixed_precision)
    return torch.autocast("cuda", dtype=dt, enabled=(dt is not None and D.cuda))


def _loss_fn(args):
    if args.loss == "BCE":
        return nn.BCEWithLogitsLoss()
    if args.loss == "CE":
        return nn.CrossEntropyLoss()
    raise ValueError("loss function not defined")


def _loss(loss_fn, out, labels):
    if isinstance(loss_fn, nn.CrossEntropyLoss):
        return loss_fn(out, torch.argmax(labels.long(), dim=1))
    return loss_fn(out, labels)


def _gather_order(n_steps, batch_size, per_rank, world):
    """Sampler position (0 .. world*per_rank-1, positions >= len(dataset) are padding) of every row of the concatenated
    per-step gathers: step s, rank r, row j holds sample number (s*batch_size + j) of rank r, i.e. position (.)*world + r."""
    rows = []
    for s_ in range(n_steps):
        n_here = min(batch_size, per_rank - s_ * batch_size)
        for r in range(world):
            j = np.arange(n_here)
            rows.append((s_ * batch_size + j) * world + r)
    return np.concatenate(rows) if rows else np.zeros(0, np.int64)


def validate(model, loader, frontend, args, D, epoch, save_pred=True):
    """TT:238-307.  NOTE the reference feeds SIGMOID outputs to the loss here (TT:266-274); kept, so valid_loss is
    comparable with its logs.  valid_loss is the mean of the per-batch losses of ALL gathered batches, the DistributedSampler's padding
    duplicates included -- exactly the reference's number; the mAP / AUC below are computed on the de-duplicated rows (a documented
    divergence: the reference keeps the duplicates there too), so the two do not describe quite the same sample set under DDP."""
    from .stats import calculate_stats
    model.eval()
    loss_fn = _loss_fn(args)
    preds, tgts, losses = [], [], []
    with torch.no_grad():
        for wave, n_valid, labels, _ in loader:
            wave, labels = wave.to(D.device, non_blocking=True), labels.to(D.device, non_blocking=True)
            with _autocast(args, D):
                x, fe = frontend(wave, n_valid.to(D.device))
                out = model(x, frontend=fe)
            out = out.float()
            if args.if_nan2num:
                out = torch.nan_to_num(out)
            out = torch.sigmoid(out)
            loss = _loss(loss_fn, out, labels)
            # ragged last batches: pad to the loader's batch size so all_gather shapes agree, mark real rows
            bs = loader.batch_size
            keep = torch.zeros(bs, dtype=torch.bool, device=D.device)
            keep[:out.shape[0]] = True
            pad = lambda t: torch.cat([t, t.new_zeros((bs - t.shape[0],) + t.shape[1:])])
            k = D.gather(keep)
            preds.append(D.gather(pad(out))[k].cpu())
            tgts.append(D.gather(pad(labels))[k].cpu())
            losses.append(D.gather(loss.detach()).cpu())
    D.barrier()
    if not D.main:
        return None, None
    output, target = torch.cat(preds).numpy(), torch.cat(tgts).numpy()
    # DistributedSampler pads every rank to the same length by repeating leading samples; the reference's plain gather keeps
    # those duplicates in the metrics (TT:285-287).  They skew mAP/AUC on small evaluation sets, so only len(dataset) rows count
    # here (rank-interleaved order restored first).
    n_data = len(loader.dataset)
    if D.world > 1 and output.shape[0] > n_data and isinstance(loader.sampler, torch.utils.data.distributed.DistributedSampler):
        per = output.shape[0] // D.world
        # batches were gathered rank-major per step; the sampler deals indices rank-minor: recover dataset order
        order = _gather_order(len(loader), loader.batch_size, per, D.world)
        keep_rows = order < n_data
        output, target = output[keep_rows], target[keep_rows]
    stats = calculate_stats(output, target)
    if save_pred:
        pdir = os.path.join(args.exp_dir, "predictions")
        if not os.path.exists(pdir):
            os.mkdir(pdir)
            np.savetxt(os.path.join(pdir, "target.csv"), target, delimiter=",")
        np.savetxt(os.path.join(pdir, f"predictions_{epoch}.csv"), output, delimiter=",")
    return stats, float(torch.cat(losses).mean())


def train(model, train_loader, val_loader, args, D):
    """TT:15-236"""
    from .stats import summarize
    model = model.to(D.device)
    trainables = [p for p in model.parameters() if p.requires_grad]
    D.print("Total parameter number is : {:.3f} million".format(sum(p.numel() for p in model.parameters()) / 1e6))
    D.print("Total trainable parameter number is : {:.3f} million".format(sum(p.numel() for p in trainables) / 1e6))
    k = args.bs_scale_factor
    if args.optim == "adam":
        optimizer = torch.optim.Adam(trainables, args.lr, weight_decay=args.weight_decay,
                                     betas=(1 - (1 - 0.95) * k, 1 - (1 - 0.999) * k), eps=1e-8 / (k ** 0.5),
                                     fused=True if D.cuda else None)     # one multi-tensor launch, step counters on the device
    else:
        optimizer = torch.optim.SGD(trainables, args.lr, momentum=0.9, weight_decay=args.weight_decay)
    if args.optim_path:
        optimizer.load_state_dict(torch.load(args.optim_path, map_location="cpu"))
    net, homes = model, None
    if D.ddp:
        net = nn.parallel.DistributedDataParallel(model, device_ids=[D.dev_index] if D.cuda else None,
                                                  gradient_as_bucket_view=True, static_graph=bool(args.if_nan2num))
        # (torch's default buckets -- 25 MB behind a first one of 1 MB: with a custom cap the odd-sized head.bias sits at the front of a big
        # bucket and leaves every view behind it off a 16-byte boundary, which the in-bucket gradient writes of ssi.grad_home need)
        homes = compress_gradients(net, args.grad_compress)
    scaler = torch.amp.GradScaler("cuda", enabled=(args.mixed_precision == "fp16" and D.cuda))
    scheduler = torch.optim.lr_scheduler.MultiStepLR(
        optimizer, list(range(args.lrscheduler_start, 1000, args.lrscheduler_step)), gamma=args.lrscheduler_decay)
    loss_fn = _loss_fn(args)
    fe_train, fe_val = Frontend(args, D.device, True), Frontend(args, D.device, False)
    D.print("now training with {:s}, main metrics: {:s}, loss function: {:s}, learning rate scheduler: {:s}".format(
        str(args.dataset), str(args.metrics), str(loss_fn), str(scheduler)))

    progress, result = [], np.zeros([args.n_epochs, 8])
    best_epoch, best_mAP, best_acc = 0, -np.inf, -np.inf
    loss_acc = torch.zeros(2, device=D.device, dtype=torch.float32)          # [sum of loss * clips, clips] of this rank, this epoch
    global_step, epoch = 0, 1
    warm_steps, warm_every = 1000 // k, max(1, 50 // k)
    while epoch < args.n_epochs + 1:
        net.train()
        D.print("---------------")
        D.print(datetime.datetime.now())
        D.print("current #epochs=%s, #steps=%s" % (epoch, global_step))
        if hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(epoch)
        t0 = time.time()
        for i, (wave, n_valid, labels, _) in enumerate(train_loader):
            if args.max_steps and i >= args.max_steps:
                break
            wave, labels = wave.to(D.device, non_blocking=True), labels.to(D.device, non_blocking=True)
            if args.warmup and global_step <= warm_steps and global_step % warm_every == 0:
                warm_lr = (global_step / warm_steps) * args.lr
                for g in optimizer.param_groups:
                    g["lr"] = warm_lr
                D.print("warm-up learning rate is {:f}".format(warm_lr))
            with torch.no_grad():
                x, fe = fe_train(wave, n_valid.to(D.device))
            with _autocast(args, D):
                out = net(x, frontend=fe)
            loss = _loss(loss_fn, out.float(), labels)
            if args.if_nan2num:
                loss = torch.nan_to_num(loss)
            else:
                # non-default (--if_nan2num False, TT:153-164): the decision to skip or stop must be the SAME on every rank --
                # a rank that skipped alone would leave the others waiting in the gradient all-reduce -- so the finite flag is
                # reduced (MIN) first.  This is the only per-step host sync of the loop and it is off by default.
                finite = torch.isfinite(loss.detach()).to(torch.float32)
                if D.ddp:
                    dist.all_reduce(finite, op=dist.ReduceOp.MIN)
                if finite.item() == 0.0:
                    if args.if_continue_inf:
                        D.print("Loss is not finite on some rank, continuing training")
                        # the step is skipped on EVERY rank (the flag was MIN-reduced): no backward, so no gradient exchange is started
                        # -- a DDP forward without a backward is legal while static_graph is off, which is why it is off without
                        # --if_nan2num -- and nothing non-finite reaches the buckets
                        optimizer.zero_grad()
                        continue
                    D.print("Loss is not finite on some rank, stopping training")
                    sys.exit(1)
            optimizer.zero_grad(set_to_none=True)
            scaler.scale(loss).backward()
            if homes is not None:
                homes.after_backward()
            scaler.step(optimizer)
            scaler.update()
            # running loss stays on the device: no .item() and no collective per step (the reference's per-step gather + print,
            # TT:157-174, is what SURVEY 5 flags as the scaling hazard); ranks exchange it every n_print_steps and per epoch
            loss_acc[0].add_(loss.detach().float(), alpha=float(wave.shape[0]))       # scalars ride as kernel arguments: no host tensor, no copy
            loss_acc[1] += float(wave.shape[0])
            global_step += 1
            HOST_SYNCS["steps"] += 1
            if global_step % args.n_print_steps == 0:
                loss_sum, loss_cnt = _sync_loss(loss_acc, D)
                D.print("Epoch {} step {} T_Loss {:.5f} ({:.1f} clips/s)".format(
                    epoch, global_step, loss_sum / max(1, loss_cnt), loss_cnt / (time.time() - t0)))

        loss_sum, loss_cnt = _sync_loss(loss_acc, D)
        D.print("start validation")
        stats, valid_loss = validate(net, val_loader, fe_val, args, D, epoch)
        if D.main:
            s = summarize(stats, args.metrics)
            train_loss = loss_sum / max(1, loss_cnt)
            lr_now = optimizer.param_groups[0]["lr"]
            D.print(("mAP: {:.6f}" if args.metrics == "mAP" else "acc: {:.6f}").format(s["main"]))
            for name, key in (("AUC", "mAUC"), ("Avg Precision", "precision"), ("Avg Recall", "recall"), ("d_prime", "d_prime")):
                D.print("{}: {:.6f}".format(name, s[key]))
            D.print("train_loss: {:.6f}".format(train_loss))
            D.print("valid_loss: {:.6f}".format(valid_loss))
            result[epoch - 1, :] = [s["main"], s["mAUC"], s["precision"], s["recall"], s["d_prime"], train_loss, valid_loss, lr_now]
            np.savetxt(args.exp_dir + "/result.csv", result, delimiter=",")
            if s["mAP"] > best_mAP:
                best_mAP = s["mAP"]
                if args.metrics == "mAP":
                    best_epoch = epoch
            if s["acc"] > best_acc:
                best_acc = s["acc"]
                if args.metrics == "acc":
                    best_epoch = epoch
            sd = net.state_dict()                                     # with the `module.` prefix under DDP, as the reference saves
            if best_epoch == epoch:
                torch.save(sd, "%s/models/best_audio_model.pth" % args.exp_dir)
                torch.save(optimizer.state_dict(), "%s/models/best_optim_state.pth" % args.exp_dir)
            if args.save_model:
                torch.save(sd, "%s/models/latest_audio_model.%d.pth" % (args.exp_dir, epoch))
                torch.save(optimizer.state_dict(), "%s/models/latest_optim_state.%d.pth" % (args.exp_dir, epoch))
            D.print("Epoch-{0} lr: {1}".format(epoch, lr_now))
            with open(args.exp_dir + "/stats_" + str(epoch) + ".pickle", "wb") as h:
                pickle.dump(stats, h, protocol=pickle.HIGHEST_PROTOCOL)
            progress.append([epoch, global_step, best_epoch, best_mAP, best_acc])
            with open("%s/progress.pkl" % args.exp_dir, "wb") as f:
                pickle.dump(progress, f)
        loss_acc.zero_()
        D.barrier()
        scheduler.step()
        epoch += 1
    return model


HOST_SYNCS = {"steps": 0, "loss_syncs": 0}      # counted so that tests can assert the hot loop's host-sync budget


def _sync_loss(loss_acc, D):
    """(sum, count) of the running training loss over all ranks: one all-reduce + one device->host copy, called every
    --n-print-steps steps and at the end of an epoch -- never per step."""
    t = loss_acc.clone()
    if D.ddp:
        dist.all_reduce(t)
    HOST_SYNCS["loss_syncs"] += 1
    v = t.tolist()
    return float(v[0]), int(round(v[1]))


def evaluate(model, val_loader, args, D, tag):
    """RUN:283-324"""
    from .stats import summarize
    stats, loss = validate(model.to(D.device), val_loader, Frontend(args, D.device, False), args, D, tag)
    if not D.main:
        return None
    s = summarize(stats, args.metrics)
    D.print(("mAP: {:.6f}" if args.metrics == "mAP" else "acc: {:.6f}").format(s["main"]))
    D.print("AUC: {:.6f}".format(s["mAUC"]))
    D.print("d_prime: {:.6f}".format(s["d_prime"]))
    D.print("valid_loss: {:.6f}".format(loss))
    res = [s["main"], s["mAUC"], s["precision"], s["recall"], s["d_prime"], loss]
    np.savetxt(args.exp_dir + f"/result_{tag}.csv", res, delimiter=",")
    with open(args.exp_dir + f"/stats_{tag}.pickle", "wb") as h:
        pickle.dump(stats, h, protocol=pickle.HIGHEST_PROTOCOL)
    return s


def main(argv=None):
    args = build_parser().parse_args(argv)
    check_scope(args)
    random.seed(EXP_SEED), np.random.seed(EXP_SEED), torch.manual_seed(EXP_SEED)
    D = Dist()
    if D.cuda:
        from . import tunable
        tunable.enable(D.local_rank)
    print("I am process %s, running on %s: starting (%s)" % (os.getpid(), os.uname()[1], time.asctime()))
    model = build_model(args)            # same initial weights on every rank (EXP_SEED); DDP broadcasts rank 0's anyway
    # from here on every rank draws its OWN random stream: SpecAug masks, noise and roll come from the device generator and the
    # loader workers from the base seed below -- with one shared seed all ranks would apply identical augmentation to their
    # i-th sample every step
    random.seed(EXP_SEED + D.rank), np.random.seed(EXP_SEED + D.rank), torch.manual_seed(EXP_SEED + D.rank)
    val_loader = make_loader(args, args.data_val, False, D)
    if args.run_type == "train":
        train_loader = make_loader(args, args.data_train, True, D)
        D.print("\nCreating experiment directory: %s" % args.exp_dir)
        if D.main:
            os.makedirs("%s/models" % args.exp_dir, exist_ok=True)
            with open("%s/args.pkl" % args.exp_dir, "wb") as f:
                pickle.dump(args, f)
        D.barrier()
        D.print("Now starting training for {:d} epochs".format(args.n_epochs))
        train(model, train_loader, val_loader, args, D)
        if args.dataset == "speechcommands" and args.data_eval:           # RUN:326-375
            D.barrier()
            sd = torch.load(args.exp_dir + "/models/best_audio_model.pth", map_location="cpu")
            model.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})
            v = evaluate(model, val_loader, args, D, "valid_set")
            e = evaluate(model, make_loader(args, args.data_eval, False, D), args, D, "eval_set")
            if D.main:
                np.savetxt(args.exp_dir + "/eval_result.csv", [v["acc"], v["mAUC"], e["acc"], e["mAUC"]])
    else:
        D.print(f"Now starting evaluation on {args.dataset} dataset!")
        os.makedirs(args.exp_dir, exist_ok=True)
        evaluate(model, val_loader, args, D, "eval")
    if D.ddp and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
