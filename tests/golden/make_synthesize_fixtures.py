#!/usr/bin/env python3
"""Generates tests/golden/synthesize/*.json by running the REFERENCE's workload
generator, bin/synthesize.py, unmodified (it needs only click and numpy), as a
child process with the argument lists of CASES.  The files are the programs it
wrote, byte for byte; tests/test_frontend.py::test_synthesize_reproduces_the_
reference_generator requires stencilflow_amd.programs.synthesize to produce the
same file name and the same JSON for the same arguments.

Runs only in the build container; the fixtures are data (generated programs),
nothing of the generator's source is stored.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "synthesize")

# positional: data_type num_stages num_fields_spatial size_x size_y size_z extent_x extent_y extent_z
CASES = [
    ["float32", "3", "0", "16", "16", "32", "1", "1", "1"],                                   # cross 3-D
    ["float64", "2", "0", "8", "8", "16", "2", "1", "1"],                                     # wider extent
    ["float32", "4", "0", "64", "64", "0", "1", "1", "0"],                                    # cross 2-D
    ["float32", "2", "0", "32", "0", "0", "2", "0", "0"],                                     # 1-D
    ["float32", "2", "0", "8", "8", "8", "1", "1", "1", "-stencil_shape", "box"],             # 27-point box
    ["float64", "4", "0.5", "64", "64", "0", "2", "1", "0", "-fork_frequency", "0.5",
     "-stencil_shape", "box"],                                                                # forks + fractional fields, 2-D box
    ["float32", "3", "1", "16", "16", "16", "1", "1", "1"],                                   # one extra field per stage
    ["float32", "5", "1.5", "16", "16", "16", "1", "1", "1", "-fork_frequency", "0.34",
     "-fork_length_left", "1", "-fork_length_right", "3"],                                    # uneven forks, 1.5 fields
    ["float32", "2", "0", "8", "8", "8", "1", "1", "1", "-stencil_shape", "diffusion"],       # diffusion 3-D
    ["float64", "3", "0", "16", "32", "0", "1", "1", "0", "-stencil_shape", "diffusion"],     # diffusion 2-D
    ["float32", "3", "0", "16", "16", "16", "1", "1", "1", "-stencil_shape", "hotspot"],      # hotspot 3-D
    ["float32", "4", "0", "32", "32", "0", "1", "1", "0", "-stencil_shape", "hotspot"],       # hotspot 2-D
    ["float32", "2", "1", "16", "16", "16", "1", "1", "1", "-stencil_shape", "hotspot"],      # hotspot, own power field per stage
    ["float32", "2", "0", "16", "16", "32", "1", "1", "1", "-vectorize", "4"],                # vectorize
    ["float32", "1000", "0", "512", "512", "512", "1", "1", "1"],                             # the C3 generator call of SURVEY §8(d)
]


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    index = []
    with tempfile.TemporaryDirectory() as tmp:
        for case in CASES:
            before = set(os.listdir(tmp))
            subprocess.run([sys.executable, os.path.join(REFERENCE, "bin", "synthesize.py")] + case, cwd=tmp,
                           check=True, stdout=subprocess.DEVNULL)
            (name, ) = set(os.listdir(tmp)) - before
            big = os.path.getsize(os.path.join(tmp, name)) > 64 * 1024
            if big:
                # 1000-stage program (~330 KB): keep name, size, and a digest instead of the file
                import hashlib
                with open(os.path.join(tmp, name), "rb") as f:
                    blob = f.read()
                index.append({"args": case, "file": name, "bytes": len(blob),
                              "sha256": hashlib.sha256(blob).hexdigest()})
            else:
                shutil.copy(os.path.join(tmp, name), os.path.join(OUT, name))
                index.append({"args": case, "file": name})
    with open(os.path.join(OUT, "index.json"), "w") as f:
        json.dump(index, f, indent=1)
    print("wrote", len(index), "programs to", OUT)


if __name__ == "__main__":
    main()
