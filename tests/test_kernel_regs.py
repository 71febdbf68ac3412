"""The headline path's kernels keep their register counts (no GPU needed: the counts are read from the code object in smatrix.so).

Round 6 found out why this has to be a test: code shared with the dense-id mechanisms (rest_enter inlined into grow_lds_task, an
eight-window scan in grow_map_body) took the SCRAMBLED stream's k_grow_lds from 18 to 93-96 registers and k_grow_map from 18 to 48,
and the frozen growth round lost 0.035 ms per step before an A/B against the round-5 tree showed it (profiles/r06_ab_vs_r05.txt).
The bounds are the counts of the build the driver's BENCH figures come from, with a few registers of slack for compiler noise."""
import os, re, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "libsmatrix_amd", "lib", "smatrix.so")

# kernel (demangled, as tools/kernel_regs.py prints it) -> (max VGPRs, max spilled VGPRs)
BOUNDS = {
    "smx::k_apply_agg<2, 1u, true, false>": (64, 0),     # the folding kernel of the headline: two 1024-lane workgroups per CU need <= 64
    "smx::k_apply<0, false>": (48, 0),                   # GET, lane per op
    "smx::k_grow_lds<64, 8u, false>": (24, 0),           # the scrambled stream's in-LDS rehashes (round 5: 18 / 18 / 21)
    "smx::k_grow_lds<256, 11u, false>": (24, 0),
    "smx::k_grow_lds<1024, 13u, false>": (28, 0),
    "smx::k_grow_map<false>": (20, 0),
    "smx::k_grow_move": (28, 0),
    "smx::k_getrow<2, true, 0>": (56, 0),               # the config-3 scan
    "smx::k_get_clu": (64, 0),                           # clustered GET: 8 waves per SIMD
}


def test_headline_kernels_keep_their_registers():
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_regs.py"), LIB], capture_output=True, text=True, timeout=600).stdout
    seen = {}
    for line in out.splitlines():
        m = re.match(r"(?:void )?(\S.*?)\s+sgpr\s+(\d+) \(spilled\s+(\d+)\)\s+vgpr\s+(\d+) \(spilled (\d+)\)", line)
        if m:
            seen[m.group(1).strip()] = (int(m.group(4)), int(m.group(5)))
    assert len(seen) > 50, "tools/kernel_regs.py found no kernels:\n" + out[:500]
    for name, (max_v, max_spill) in BOUNDS.items():
        assert name in seen, "kernel %s is not in the library (renamed? update this test and DESIGN 3.1)" % name
        v, sp = seen[name]
        assert v <= max_v and sp <= max_spill, "%s: %d VGPRs (%d spilled), the bound is %d (%d)" % (name, v, sp, max_v, max_spill)
