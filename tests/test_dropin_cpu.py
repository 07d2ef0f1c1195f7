"""The drop-in boundary driven by the reference's own factory and optimizer builder (oracle/check_dropin.py; CPU,
construction only).  Needs /root/reference, which exists in the build container only: skipped elsewhere."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MADRIGAL_REFERENCE", "/root/reference")


def test_adaptor_and_encoder_are_sibling_classes():
    """The reference's create_optimizer (madrigal/utils.py:467-479) tells the fusion-side adaptors from the tabular encoders
    with isinstance(): MLPAdaptor must not be an MLPEncoder (nor the other way round), as in the reference (models.py:121, :459)."""
    from madrigal_amd import models as M
    assert not issubclass(M.MLPAdaptor, M.MLPEncoder) and not issubclass(M.MLPEncoder, M.MLPAdaptor)
    a, e = M.MLPAdaptor(8, [8], 4, 0.0, "ln", "relu"), M.MLPEncoder(8, [8], 4, 0.0, None, "relu")
    assert sorted(a.state_dict()) == sorted(e.state_dict())


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "madrigal")), reason="reference tree not present (GPU box)")
def test_reference_get_model_and_create_optimizer_over_the_aliased_classes():
    r = subprocess.run([sys.executable, os.path.join(REPO, "oracle", "check_dropin.py"), "--ref", REF], capture_output=True,
                       text=True, timeout=600, env={**os.environ, "OMP_NUM_THREADS": "4"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "check_dropin ok" in r.stdout
    assert r.stdout.count("[ok]") == 3
