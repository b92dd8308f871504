# same-box A/B of the packed (bf16 pair) state checkpoints of the token-major scan against the fp32 ones (a -DAUM_SCANT_CK_F32 build from
# tools/build_variant.sh): parity tests, the two scan kernels alone, the whole step
set -x
export PYTHONPATH=$PWD/audio-mamba-aum_amd:$PYTHONPATH
mkdir -p gpurun_out/ckpt_ab
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "tm or token_major or headline or repeatable or inner_fns" 2>&1 | tail -4
V=$PWD/audio-mamba-aum_amd/aum_hip/variants/libaum_hip_ckf32.so
for i in 1 2; do
  timeout 200 python tools/kbench.py --only scan_tm 2>&1 | grep -v amdgpu | grep "scan_tm_fwd_bidir_train\|scan_tm_bwd_bidir" | sed "s/^/packed_$i /"
  AUM_DEBUG=1 AUM_HIP_LIB=$V timeout 200 python tools/kbench.py --only scan_tm 2>&1 | grep -v amdgpu | grep "scan_tm_fwd_bidir_train\|scan_tm_bwd_bidir" | sed "s/^/fp32ck_$i /"
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/ckpt_ab/bench_packed_$i.json 2>gpurun_out/ckpt_ab/bench_packed_$i.err
  AUM_DEBUG=1 AUM_HIP_LIB=$V timeout 300 python bench.py --no-cpu-baseline > gpurun_out/ckpt_ab/bench_fp32ck_$i.json 2>gpurun_out/ckpt_ab/bench_fp32ck_$i.err
done
python -c "
import json
for k in ('packed_1','fp32ck_1','packed_2','fp32ck_2'):
    d=json.load(open('gpurun_out/ckpt_ab/bench_%s.json'%k)); km=d['kernel_ms_per_step']; print(k, d['ms_per_step'], d['value'], km.get('scan_tm_fwd_bidir'), km.get('scan_tm_bwd_bidir'), d['final_loss'])
"
tail -c 1500 gpurun_out/ckpt_ab/bench_packed_1.err
