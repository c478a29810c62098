"""The opt-in pipelined weight-gradient kernel (csrc/pmlp_wgrad_pipe.h, NSVD_WGRAD_PIPE=1; DESIGN.md 3.2: correct but
slower than the tile kernel, kept as the measured record of that design) must stay correct: the trainer / backward
parity tests run again in a child interpreter with the switch set (the library reads it once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_backward_parity_suite_with_the_pipelined_weight_gradient_kernel():
    env = dict(os.environ, NSVD_WGRAD_PIPE="1")
    sel = ["tests/test_dropin_gpu.py::test_fused_trainer_step_matches_oracle",
           "tests/test_dropin_gpu.py::test_optimiser_step_fused_into_backward_is_bit_identical",
           "tests/test_hip_parity.py::test_backward_given_df_headline",
           "tests/test_hip_parity.py::test_backward_headline_size_sampled_heads",
           "tests/test_hip_parity.py::test_model_forward_backward_mfma"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", *sel], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout
