"""The weight-gradient kernel (K2, csrc/txp_wgrad_bf16.hip) on its own: tools/micro/k2_bench (built by
__graft_entry__.build()) runs the library's kernel on synthetic saved arrays and prints its distance to an fp64 host sum per
layer.  Small batches matter here: with one round per workgroup the item loop's first iteration is all there is, and that is
where an instruction-hazard bug of the kernel's inline-assembly MFMAs once lived -- invisible at bench sizes
against a tolerance, obvious against fp64 at N = 1..40."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "micro", "k2_bench")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,v,ragged,bf16", [(1, 32, 0, 0), (3, 32, 1, 0), (8, 32, 1, 0), (8, 32, 1, 1), (16, 32, 0, 1),
                                             (40, 32, 1, 0), (24, 64, 1, 0), (24, 64, 1, 1), (12, 128, 1, 0), (64, 20, 1, 0)])
def test_weight_gradient_kernel_against_fp64_host_sum(n, v, ragged, bf16):
    if not os.path.exists(BIN):
        pytest.fail("tools/micro/k2_bench is not built (__graft_entry__.build() builds it)")
    env = dict(os.environ, K2_GARBAGE="1")            # nonzero values in the channels of a_0 that no layer reads
    out = subprocess.run([BIN, str(n), str(v), str(ragged), str(bf16)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = re.findall(r"layer (\d+): shipped vs fp64 host: max \|diff\| (\S+) of (\S+)", out.stdout)
    assert len(rows) == 5, out.stdout
    for layer, diff, ref in rows:
        assert float(diff) <= 1e-6 * max(float(ref), 1e-3), (layer, diff, ref)
    # the tree's kernel source compiled into the harness (both workgroup shapes) against the library's
    for rel in re.findall(r"vs shipped: max \|diff\| \S+ of max \|ref\| \S+  \(rel (\S+),", out.stdout):
        assert float(rel) <= 2e-6, out.stdout
