"""A slab-decomposed run driven through include/sf_hip.h alone: tests/capi_slab_demo.c
(plain C: fork per rank, sf_plan_* + sf_halo_*, socket pairs for the 256-byte buffer
descriptions) against the oracle.  CPU: the program compiles and links against the
library; GPU: 2 and 3 ranks on this box's GPU, results bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import stencilflow_amd as sf
from stencilflow_amd import programs
from stencilflow_amd.lowering import lower

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stencilflow_amd", "csrc")


def _build(tmp_path):
    exe = str(tmp_path / "capi_slab_demo")
    cmd = ["gcc", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "capi_slab_demo.c"), "-L", CSRC, "-lsf_hip",
           "-Wl,-rpath," + CSRC, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_driver_of_a_decomposed_run_builds(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "usage:" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("world,fuse,stages,halo", [(2, 2, 6, 0), (3, 2, 5, 0), (3, 1, 3, 0), (3, 2, 17, 8), (2, 3, 11, 6)])
def test_c_driver_of_a_decomposed_run(tmp_path, world, fuse, stages, halo):
    """halo = 0: the per-launch loop of the demo (halo = one launch's reach); halo > 0:
    one call of the library's deep-halo schedule, sf_plan_execute_decomposed."""
    from oracle import numpy_oracle as npo
    exe = _build(tmp_path)
    shape = (16 * world + 3, 20, 64)
    prog = programs.jacobi3d(shape, stages, bc_value=0.25)
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    sfir = str(tmp_path / "p.sfir")
    with open(sfir, "w") as f:
        f.write(lower(sf.KernelChainGraph(path)))
    x = np.random.default_rng(5).uniform(-1, 1, shape).astype(np.float32)
    x.tofile(str(tmp_path / "a.dat"))
    env = dict(os.environ, SF_HIP_OPTIONS="fuse={}".format(fuse), HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = [exe, sfir, str(tmp_path / "a.dat"), str(tmp_path / "out"), str(world), str(shape[0]),
            str(shape[1] * shape[2] * 4), str(halo or fuse)] + (["deep"] if halo else [])
    r = subprocess.run(args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    want = npo.run_reference(prog, {"a": x})["b%d" % (stages - 1)]
    for rank in range(world):
        lo, hi = shape[0] * rank // world, shape[0] * (rank + 1) // world
        got = np.fromfile(str(tmp_path / ("out.%d" % rank)), np.float32).reshape((hi - lo, ) + shape[1:])
        assert np.array_equal(got, want[lo:hi]), rank


@pytest.mark.gpu
@pytest.mark.parametrize("fuse,stages", [(2, 6), (1, 3)])
def test_c_driver_over_the_rccl_rung_sends_to_itself(tmp_path, fuse, stages):
    """The RCCL rung of the library driven from plain C with no torch in the process
    (VERDICT r02, next 4): what a one-GPU box can run of it is a communicator of one
    rank -- rank 1 of 3, whose halos come back to itself (send_down lands in its lower
    ghost planes, send_up in the upper ones).  The per-launch loop then equals, bit for
    bit, the oracle applied to the slab extended by copies of its own boundary planes."""
    from oracle import numpy_oracle as npo
    exe = _build(tmp_path)
    world = 3
    shape = (16 * world + 3, 20, 64)
    prog = programs.jacobi3d(shape, stages, bc_value=0.25)
    path = programs.write_program(prog, str(tmp_path / "p.json"))
    sfir = str(tmp_path / "p.sfir")
    with open(sfir, "w") as f:
        f.write(lower(sf.KernelChainGraph(path)))
    x = np.random.default_rng(6).uniform(-1, 1, shape).astype(np.float32)
    x.tofile(str(tmp_path / "a.dat"))
    env = dict(os.environ, SF_HIP_OPTIONS="fuse={}".format(fuse), HSA_ENABLE_IPC_MODE_LEGACY="0")
    args = [exe, sfir, str(tmp_path / "a.dat"), str(tmp_path / "out"), str(world), str(shape[0]),
            str(shape[1] * shape[2] * 4), str(fuse), "rccl-self"]
    r = subprocess.run(args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lo, hi = shape[0] // world, shape[0] * 2 // world
    slab, done = x[lo:hi].copy(), 0
    while done < stages:  # one launch: `t` fused operators on the slab extended by its own boundary planes
        t = min(fuse, stages - done)
        ext = np.concatenate([slab[:t], slab, slab[-t:]])
        sub = programs.jacobi3d(ext.shape, t, bc_value=0.25)
        slab = npo.run_reference(sub, {"a": ext})["b%d" % (t - 1)][t:-t]
        done += t
    got = np.fromfile(str(tmp_path / "out.1"), np.float32).reshape(slab.shape)
    assert np.array_equal(got, slab)
