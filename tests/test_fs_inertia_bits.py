"""transform_spatial_inertia of the Featherstone kernels (newton_amd/csrc/nt_featherstone.hpp) sums only the terms of T^T I T whose
factor is not a structural zero; tools/fs_inertia_bitcheck.py compiles it for the host next to the dense 6 x 6 products the reference
performs (featherstone/kernels.py:66-139) and compares 2 000 000 random and degenerate inputs bit for bit."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sparse_spatial_inertia_transform_is_bit_identical_to_the_dense_products():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fs_inertia_bitcheck.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 differ" in r.stdout
