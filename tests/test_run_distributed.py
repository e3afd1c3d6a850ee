"""bin/run_distributed_program.py: one program on N ranks of this node (the place of
the reference's `mpirun -n N bin/run_distributed_program.py`,
bin/run_distributed_program.py:98-100,283-341), slabs along the outermost dimension.
GPU: three ranks on this box's one GPU (-single-device), results stitched by rank 0
and verified against the named CPU checker."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "bin", "run_distributed_program.py")


def _env():
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""),
             HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SF_REFERENCE_CHECKER"):
        e.pop(k, None)
    return e


def test_command_line_surface():
    r = subprocess.run([sys.executable, CLI, "--help"], capture_output=True, text=True, env=_env())
    assert r.returncode == 0
    for flag in ("-gpus", "-compare-to-reference", "-reference-checker", "-halo", "-repetitions", "-input-directory"):
        assert flag in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("program,ranks", [("jacobi3d_32x32x32_8itr_8vec", 3), ("jacobi2d_128x128", 2)])
def test_program_on_three_ranks_matches_the_checker(programs_dir, tmp_path, program, ranks):
    from oracle import numpy_oracle as npo
    path = os.path.join(programs_dir, program + ".json")
    r = subprocess.run([sys.executable, CLI, path, "hardware", "-gpus", str(ranks), "-single-device",
                        "-compare-to-reference", "-reference-checker", "tests.reference_provider:reference_outputs",
                        "-input-directory", programs_dir], cwd=str(tmp_path), capture_output=True, text=True,
                       env=_env(), timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "Results verified." in r.stdout and "slab(s)" in r.stdout
    want = npo.run_reference(path, input_directory=programs_dir)
    for name, ref in want.items():
        got = np.fromfile(str(tmp_path / "results" / program / (name + ".dat")), ref.dtype).reshape(ref.shape)
        assert np.array_equal(got, ref), name
    assert not os.path.exists(str(tmp_path / "results" / program / ".parts"))
