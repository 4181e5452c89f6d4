"""Deterministic, key-addressed random weights shared by the golden generator and the tests: re-exported from
prediff_amd.seeding (bench.py and __graft_entry__.smoke() use the same functions without importing the test tree)."""
from prediff_amd.seeding import heavy_tailed_state_dict, seeded_input, seeded_state_dict, seeded_tensor  # noqa: F401
