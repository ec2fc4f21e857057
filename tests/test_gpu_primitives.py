"""Hardware assumptions of every kernel (MFMA operand layout, ds_read_b64_tr_b16 mapping)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_probe_primitives():
    exe = os.path.join(ROOT, "efficient-attention_amd", "lib", "probe_primitives")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "PROBE ALL OK" in r.stdout, r.stdout
