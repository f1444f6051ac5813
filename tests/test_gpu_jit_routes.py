"""on the GPU box, in a process that has ALREADY initialised the GPU and launched kernels, generated
kernels are still compiled by the helper process (`gjx_jitc`, a fresh child started with posix_spawn — the caller is never
replaced) and not by hiprtc inside this process; a compiler that dies there is GJX_ERR_JIT for the caller.  `gjx_jit_routes`
is the evidence: child_compiles grows, inproc_compiles and spawn_failures stay 0."""

import os

import pytest
import torch

from genjax._amd import abi, prng, workloads as W

pytestmark = pytest.mark.gpu


def _fresh_plan(ops, scale):
    sites = W.gaussian10_sites(W.gaussian10_data())[:4]
    sites[1].arg[1] = abi.Arg(abi.ARG_CONST, 0, 0.0, scale, None)  # another constant = another kernel source
    return ops.plan_create(sites)


def test_generated_kernels_are_compiled_out_of_process(hip_ops, oracle_ops):
    assert not os.environ.get("GJX_JIT_INPROC") and not os.environ.get("GJX_JIT_INPROC_FALLBACK")
    n = 4096
    x = torch.randn(n, device="cuda")
    torch.cuda.synchronize()  # this process owns an initialised GPU and has launched kernels
    kb = W.importance_particle_keys(prng.key(3, 1), n)
    warm = _fresh_plan(hip_ops, 0.73125)
    hip_ops.importance_run(warm, kb, n, [], [torch.float32, torch.float32])
    torch.cuda.synchronize()
    r0, s0 = hip_ops.jit_routes(), hip_ops.jit_stats()
    plan = _fresh_plan(hip_ops, 0.73126 + 1e-3 * (os.getpid() % 97))  # a structure this process has not compiled
    vals, score, logw, _ = hip_ops.importance_run(plan, kb, n, [], [torch.float32, torch.float32])
    torch.cuda.synchronize()
    r1, s1 = hip_ops.jit_routes(), hip_ops.jit_stats()
    assert s1["compiles"] == s0["compiles"] + 1
    assert r1["child_compiles"] == r0["child_compiles"] + 1, (r0, r1)
    assert r1["inproc_compiles"] == 0 and r1["spawn_failures"] == 0, r1
    # ... and what the child compiled is the right kernel: bit-equal to the oracle
    oplan = _fresh_plan(oracle_ops, 0.73126 + 1e-3 * (os.getpid() % 97))
    _, _, ologw, _ = oracle_ops.importance_run(oplan, kb, n, [], [torch.float32, torch.float32])
    assert torch.equal(logw.cpu(), ologw)
    del x


def test_a_compiler_crash_on_the_gpu_box_is_an_error_code(hip_ops):
    torch.zeros(8, device="cuda").sum().item()  # GPU initialised, kernels launched
    r0 = hip_ops.jit_routes()
    good = b'#include "gjx_device.hpp"\nextern "C" __global__ void k(float* x) { x[threadIdx.x] = gjx::u2f(0x3f800000u); }\n'
    crash = b'#include "gjx_device.hpp"\n#pragma clang __debug crash\nextern "C" __global__ void k(float* x) { x[0] = 1.0f; }\n'
    assert hip_ops.lib._gjx_jit_compile_source(crash) == -6  # GJX_ERR_JIT, and this process is still here
    assert hip_ops.lib._gjx_jit_compile_source(good) == 0
    r1 = hip_ops.jit_routes()
    assert r1["child_failures"] == r0["child_failures"] + 1 and r1["child_compiles"] == r0["child_compiles"] + 1
    assert r1["inproc_compiles"] == 0 and r1["spawn_failures"] == 0
    # the device is still usable by this process afterwards
    assert float(torch.ones(1000, device="cuda").sum().item()) == 1000.0
