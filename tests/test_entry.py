"""The driver's build check: `__graft_entry__.build()` compiles the library, plans every kernel family at benchmark size
and asserts which family each plan lands on -- expectations that move when the planner's defaults move (round 4: the
generator's boxes went from the compact kernel to the dense kernel's fused form and nothing ran this function until the
end of the round)."""
import importlib


def test_build_entry_point_runs():
    entry = importlib.import_module("__graft_entry__")
    entry.build()
