"""The builds' radix sort (dxrvoxelizer_amd/csrc/radix_sort.hip) against std::stable_sort on the sorted field: tools/micro/sort_check
is the product's source file compiled into a test program (__graft_entry__.build() makes it); every size class (1 key to 6.4 M), both
callers' fields (the LBVH's 30 bits of Morton code, the lists' texel | far radius), every plan the option sortbits can ask for."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_radix_sort_equals_stable_sort_for_every_plan_and_size():
    exe = os.path.join(ROOT, "tools", "micro", "sort_check")
    assert os.path.exists(exe), "tools/micro/sort_check is missing: python -c 'import __graft_entry__ as g; g.build()'"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    checked = [r for r in rows if "differences" in r]
    assert len(checked) >= 60 and all(r["differences"] == 0 for r in checked), [r for r in checked if r["differences"]][:3]
    assert {r["n"] for r in checked} >= {1, 63, 4097, 70000, 1000000, 6403636} and {r["passes"] for r in checked} >= {3, 4, 5}
