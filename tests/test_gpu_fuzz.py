"""GPU: a short run of the two fuzzers (scripts/gpu_fuzz_routes.py, scripts/gpu_fuzz_steps.py): random maps, scans and particle
clouds -- single observes, and runs of whole steps with resampling -- through the production routes (k_step_pub, k_step_pub_big)
against the general kernels, maps bit for bit.  The long runs (6 000 scenes, 800 runs) are in profiles/r04/."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def test_random_scenes_through_the_production_routes(lib):
    from gpu_fuzz_routes import fuzz_routes

    assert fuzz_routes(80, 21, lib) == 0


def test_random_runs_of_whole_steps_through_the_production_routes(lib):
    from gpu_fuzz_steps import fuzz_steps

    assert fuzz_steps(10, 22, lib) == 0
