"""TEST INFRASTRUCTURE ONLY - re-export of the seeded synthetic weight / input generators.

They live in ``fullsubnet_plus_amd/synthetic.py`` (data generation only, also used by bench.py, which must not reach into
``oracle/`` outside its cpu_baseline leg); the golden fixtures of tests/golden were generated with exactly these functions."""
from fullsubnet_plus_amd.synthetic import (  # noqa: F401
    make_inputs, make_state_dict, make_state_dict_fullsubnet, make_wave,
)
