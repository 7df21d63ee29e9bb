"""Register / scratch budgets of the hot kernels, from the compiler's own resource report (no GPU needed: hipcc cross-compiles).

Round 3 found three product kernels taxed by optional features that had been added as run-time branches ("Bugs found in
round 3": the LN kernel's phase stamps and mm8 hooks, the fused WKV7 kernel's mm8 prologue) -- each visible in this report as a
register count that crossed an occupancy step or as spilled words, none visible in any parity test.  The budgets below are the
values of the lean instantiations plus a little room; a change that exceeds one should become a template form of its own."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "chirrup_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Rpass-analysis=kernel-resource-usage", "-c"]

# (file, substring of the mangled kernel name) -> limits
BUDGET = {
    ("wkv7.hip", "wkv7_seq_kernelILi0ELb0E"): dict(vgprs=128, scratch=0, waves=4),          # the reference's operator
    ("wkv7.hip", "wkv7_seq_kernelILi1ELb0E"): dict(vgprs=128, scratch=0, waves=4),          # fused decode form, binary16 model
    ("wkv7.hip", "wkv7_seq_kernelILi2ELb0E"): dict(vgprs=128, scratch=16, waves=4),         # ... with the mm8 prologue
    ("elementwise.hip", "add_ln_mix_kernelILi6ELi1024ELb0ELb0E"): dict(vgprs=112, scratch=0),   # LN1, binary16 model
    ("elementwise.hip", "add_ln_mix_kernelILi1ELi1024ELb0ELb0E"): dict(vgprs=100, scratch=0),   # LN2
    ("elementwise.hip", "add_ln_mix_kernelILi6ELi1024ELb0ELb1E"): dict(vgprs=128, scratch=0),   # LN1 with the mm8 hooks (1024 lanes: 128 is the limit)
    ("elementwise.hip", "add_ln_mix_kernelILi1ELi1024ELb0ELb1E"): dict(vgprs=128, scratch=0),
    ("sampler.hip", "sample_topp_kernel"): dict(vgprs=128, scratch=16),                      # 64 masses + 32 packed key pairs per lane
    ("sampler.hip", "penalize_argmax_kernel"): dict(vgprs=64, scratch=0),
}


def _report(src):
    out = subprocess.run([HIPCC] + FLAGS + [os.path.join(CSRC, src), "-o", os.devnull], capture_output=True, text=True, cwd=CSRC)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        for key, pat in (("vgprs", r"\bVGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("waves", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    return kernels


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", sorted({f for f, _ in BUDGET}))
def test_hot_kernels_stay_inside_their_register_budgets(src):
    kernels = _report(src)
    for (f, sub), lim in BUDGET.items():
        if f != src:
            continue
        hits = {n: r for n, r in kernels.items() if sub in n}
        assert hits, f"{src}: no kernel matching {sub} (renamed? update the table)"
        for name, r in hits.items():
            assert r["vgprs"] <= lim["vgprs"], (name, r, lim)
            assert r["scratch"] <= lim["scratch"], (name, r, lim)
            if "waves" in lim:
                assert r["waves"] >= lim["waves"], (name, r, lim)
