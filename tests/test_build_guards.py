"""CPU guards on the shipped binary and on the device code (no GPU needed: hipcc cross-compiles gfx950).

1. The prebuilt libspgemm_hip.so travels to the GPU box with the snapshot; its .buildinfo records the sha256 of the sources
   it was built from.  The test rebuilds (make is a no-op when nothing changed) and then requires that those hashes equal
   the sources in the tree -- a stale .so cannot be what the GPU tests load.  It runs on the GPU box as well.
2. Round 2 found a lost-update race: a return-less LDS atomic (ds_or / ds_add_f32) still in flight when its wave passed
   s_barrier, because nothing waits for an LDS op whose result nobody reads.  The fix was an explicit lgkmcnt(0) wait; the
   other ~75 block barriers rely on the compiler for the same wait.  The guard compiles the device code to ISA and fails
   if any s_barrier can be reached from an LDS instruction of its basic block without an `s_waitcnt ... lgkmcnt(0)` in
   between.
"""
import hashlib
import os
import re
import subprocess

import pytest

from helpers import ROOT

CSRC = os.path.join(ROOT, "sparse_matrix_with_flops_amd", "csrc")
SO = os.path.join(ROOT, "sparse_matrix_with_flops_amd", "libspgemm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def test_shipped_library_matches_sources():
    import __graft_entry__ as ge
    ge.build()
    info = SO + ".buildinfo"
    assert os.path.exists(SO) and os.path.exists(info), "libspgemm_hip.so / .buildinfo missing after build()"
    lines = open(info).read().splitlines()
    start = lines.index("sources sha256:")
    seen = {}
    for ln in lines[start + 1:]:
        h, name = ln.split()
        seen[name] = h
    assert set(seen) >= {"spgemm_hip.hip", "spgemm_device.hpp", "coo_device.hpp", "../../include/spgemm_hip.h"}, seen
    for name, h in seen.items():
        assert _sha(os.path.normpath(os.path.join(CSRC, name))) == h, f"{name} changed after libspgemm_hip.so was built"
    assert os.path.getmtime(SO) >= max(os.path.getmtime(os.path.normpath(os.path.join(CSRC, n))) for n in seen), \
        "libspgemm_hip.so is older than its sources"


# ---- ISA guard -------------------------------------------------------------------------------------------------------
_LABEL = re.compile(r"^\s*[.\w$]+:\s*(;.*)?$")
_BRANCH = re.compile(r"^\s*(s_cbranch|s_branch|s_setpc|s_endpgm|s_swappc)")
_DS = re.compile(r"^\s*ds_")
_WAIT = re.compile(r"^\s*s_waitcnt\b(.*)$")


def _lgkm_zero(args):
    """does this s_waitcnt wait for lgkmcnt(0)?  (symbolic form `lgkmcnt(0)` or a raw immediate with the field = 0)"""
    if "lgkmcnt(0)" in args:
        return True
    m = re.match(r"^\s*(0x[0-9a-fA-F]+|\d+)\s*(;.*)?$", args)
    if m:
        imm = int(m.group(1), 0)
        return ((imm >> 8) & 0xF) == 0           # gfx9 encoding: lgkmcnt = bits 11:8
    return False


def barriers_without_lds_wait(asm_text):
    """-> list of (function, line number) of s_barrier instructions preceded, inside their basic block, by an LDS
    instruction with no lgkmcnt(0) wait between the two."""
    bad = []
    func = "?"
    pending_ds = None                            # line of the latest un-waited LDS op on the fall-through path
    fallthrough = False                          # can control reach the next line from the previous one?
    for no, ln in enumerate(asm_text.splitlines(), 1):
        code = ln.split(";")[0].rstrip()
        if not code.strip():
            continue
        if _LABEL.match(ln):
            name = code.strip()[:-1]
            if not name.startswith(".L"):
                func = name
                pending_ds = None
            if not fallthrough:                  # entered only by branches: judged from the branch sites' own blocks
                pending_ds = None                # (a pending LDS op of the fall-through path stays pending: conservative)
            continue
        fallthrough = True
        if _DS.match(code):
            pending_ds = no
            continue
        w = _WAIT.match(code)
        if w and _lgkm_zero(w.group(1)):
            pending_ds = None
            continue
        if code.strip().startswith("s_barrier"):
            if pending_ds is not None:
                bad.append((func, no, pending_ds))
            continue
        if re.match(r"^\s*(s_branch|s_setpc|s_endpgm|s_swappc)", code):
            fallthrough = False                  # unconditional: the next line is reached only through its label
    return bad


def test_isa_guard_detects_a_missing_wait():
    good = "f:\n  ds_or_b32 v1, v2\n  s_waitcnt lgkmcnt(0)\n  s_barrier\n"
    bad = "f:\n  ds_or_b32 v1, v2\n  s_waitcnt vmcnt(0)\n  s_barrier\n"
    raw = "f:\n  ds_add_f32 v1, v2\n  s_waitcnt 0xc07f\n  s_barrier\n"
    assert barriers_without_lds_wait(good) == []
    assert barriers_without_lds_wait(raw) == []
    assert [b[0] for b in barriers_without_lds_wait(bad)] == ["f"]


@pytest.mark.timeout(600)
def test_no_barrier_leaves_an_lds_op_in_flight(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path / "spgemm_device.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-S",
           "-w", "-o", str(out), os.path.join(CSRC, "spgemm_hip.hip")]
    subprocess.check_call(cmd)
    text = out.read_text()
    nbar = sum(1 for ln in text.splitlines() if ln.split(";")[0].strip().startswith("s_barrier"))
    assert nbar > 50, f"only {nbar} s_barrier instructions found: is this the device code?"
    bad = barriers_without_lds_wait(text)
    assert not bad, "s_barrier reachable with an LDS op in flight (function, barrier line, LDS op line): " + repr(bad[:10])
