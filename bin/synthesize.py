#!/usr/bin/env python3
"""Synthetic stencil-program generator; same arguments and file naming as the
reference's bin/synthesize.py (:34-61)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stencilflow_amd.programs import synthesize  # noqa: E402

if __name__ == "__main__":
    p = argparse.ArgumentParser(description=__doc__)
    p.add_argument("data_type", choices=["float32", "float64"])
    p.add_argument("num_stages", type=int)
    p.add_argument("num_fields_spatial", type=float,
                   help="input fields per stencil read from memory (fractions allowed)")
    for n in ("size_x", "size_y", "size_z", "extent_x", "extent_y", "extent_z"):
        p.add_argument(n, type=int)
    p.add_argument("-fork_frequency", type=float, default=0.0)
    p.add_argument("-fork_length_left", type=int, default=2)
    p.add_argument("-fork_length_right", type=int, default=2)
    p.add_argument("-stencil_shape", choices=["cross", "box", "diffusion", "hotspot"], default="cross")
    p.add_argument("-vectorize", type=int, default=1)
    a = p.parse_args()
    program, filename = synthesize(**vars(a))
    with open(filename, "w") as f:
        f.write(json.dumps(program, indent=True))
    print("Wrote synthetic stencil to: {}".format(filename))
