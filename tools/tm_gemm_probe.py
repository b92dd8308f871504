#!/usr/bin/env python3
"""Library-GEMM times of the token-major block's projections at the AuM-Base shape (B*L = 64*513 tokens, bf16): what torch.matmul
(hipBLASLt) does for the row-major forms before any hand-written kernel replaces them.  Cold-ish: operands rotate through 4 buffers."""
import torch

dev, dt = "cuda", torch.bfloat16
T, Dm, E, R, N = 64 * 513, 768, 1536, 48, 16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
NB = 4
h = [rnd(T, Dm) for _ in range(NB)]
conv = [rnd(T, E) for _ in range(NB)]
xdbl = [rnd(T, R + 2 * N) for _ in range(NB)]
dxz = [rnd(T, 2 * E) for _ in range(NB)]
dout = [rnd(T, Dm) for _ in range(NB)]
W_in, W_x, W_dt, W_out = rnd(2 * E, Dm), rnd(R + 2 * N, E), rnd(E, R), rnd(Dm, E)


def timeit(name, fn, flops, byts):
    for i in range(3):
        fn(i % NB)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for i in range(n):
        fn(i % NB)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name:46s} {us:8.1f} us   {flops / us / 1e6:7.1f} TFLOP/s   {byts / us / 1e3:7.1f} GB/s")


mm = torch.matmul
timeit("in_proj fwd  h[T,768] @ W^T -> [T,3072]", lambda i: mm(h[i], W_in.t()), 2 * T * Dm * 2 * E, 2 * (T * Dm + T * 2 * E))
timeit("in_proj dgrad dxz[T,3072] @ W -> [T,768]", lambda i: mm(dxz[i], W_in), 2 * T * Dm * 2 * E, 2 * (T * Dm + T * 2 * E))
timeit("in_proj wgrad dxz^T @ h -> [3072,768]", lambda i: mm(dxz[i].t(), h[i]), 2 * T * Dm * 2 * E, 2 * (T * Dm + T * 2 * E))
timeit("out_proj fwd y[T,1536] @ W^T -> [T,768]", lambda i: mm(conv[i], W_out.t()), 2 * T * Dm * E, 2 * (T * Dm + T * E))
timeit("out_proj dgrad dout[T,768] @ W -> [T,1536]", lambda i: mm(dout[i], W_out), 2 * T * Dm * E, 2 * (T * Dm + T * E))
timeit("out_proj wgrad dout^T @ y -> [768,1536]", lambda i: mm(dout[i].t(), conv[i]), 2 * T * Dm * E, 2 * (T * Dm + T * E))
timeit("x_proj fwd conv[T,1536] @ Wx^T -> [T,80]", lambda i: mm(conv[i], W_x.t()), 2 * T * E * 80, 2 * (T * E + T * 80))
timeit("dt_proj fwd xdbl[:, :48] @ Wdt^T -> [T,1536]", lambda i: mm(xdbl[i][:, :R], W_dt.t()), 2 * T * E * R, 2 * (T * E + T * R))
timeit("dt_proj dgrad ddelta[T,1536] @ Wdt -> [T,48]", lambda i: mm(conv[i], W_dt), 2 * T * E * R, 2 * (T * E + T * R))
timeit("dt_proj wgrad ddelta^T @ xdbl[:, :48] -> [1536,48]", lambda i: mm(conv[i].t(), xdbl[i][:, :R]), 2 * T * E * R, 2 * (T * E + T * R))
timeit("x_proj wgrad dxdbl^T @ conv -> [80,1536]", lambda i: mm(xdbl[i].t(), conv[i]), 2 * T * E * 80, 2 * (T * E + T * 80))
timeit("x_proj dgrad dxdbl[T,80] @ Wx -> [T,1536]", lambda i: mm(xdbl[i], W_x), 2 * T * E * 80, 2 * (T * E + T * 80))
acc = [rnd(T, E) for _ in range(NB)]
timeit("x_proj dgrad as addmm into du", lambda i: acc[i].addmm_(xdbl[i], W_x), 2 * T * E * 80, 2 * (2 * T * E + T * 80))
