"""CPU-side checks of the C-ABI library: it builds (cross-compiles for gfx950 without a GPU), loads,
exports every symbol include/biokanga_amd.h declares, and FAILS LOUDLY (no CPU fallback) when no
HIP device is present.  No compute calls here."""
import ctypes
import os
import re
import subprocess

import pytest

import helpers

ROOT = helpers.ROOT


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "biokanga_amd", "lib", "libbiokanga_amd.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "biokanga_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)
    import biokanga_amd
    return biokanga_amd.load_library()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "biokanga_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(bk_[a-z_0-9]+)\s*\(", hdr)))


def test_exports_every_declared_symbol(lib):
    from biokanga_amd.binding import EXPORTED_SYMBOLS
    decl = declared_symbols()
    assert len(decl) >= 20
    assert sorted(EXPORTED_SYMBOLS) == decl
    for s in decl:
        assert getattr(lib, s) is not None
    assert b"biokanga_amd" in lib.bk_version()


def test_struct_layouts_match_header():
    import biokanga_amd as bk
    from biokanga_amd import binding
    assert bk.HIT_DTYPE.itemsize == 20                      # bk_hit
    assert ctypes.sizeof(bk.AlignParams) == 48              # bk_align_params
    assert bk.ENTRY_DTYPE.itemsize == 112                   # bk_entry_info (8-byte aligned)
    assert ctypes.sizeof(binding._Counters) == 64
    assert ctypes.sizeof(binding._Timing) == 52              # bk_timing: 5 + 4 floats, 4 launch counts
    assert helpers.HIT_DTYPE == bk.HIT_DTYPE                # oracle ora_hit has the same layout


def test_no_cpu_fallback(lib, golden_tmp):
    """Without a HIP device the library refuses to work instead of computing on the CPU."""
    import biokanga_amd as bk
    if bk.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(bk.BkError) as e:
        bk.Aligner(os.path.join(golden_tmp["basic"], "genome.sfx"), bk.AlignParams(max_subs=3))
    assert e.value.rc == -2                                  # BK_ERR_NODEVICE
    assert b"no CPU fallback" in lib.bk_strerror(-2)
    with pytest.raises(bk.BkError):
        bk.build_sa_device(1, 10, 1, 4, 0)


def test_product_never_touches_oracle():
    """oracle/ is test infrastructure: nothing under biokanga_amd/ or include/ may reference it."""
    bad = []
    for base in ("biokanga_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if "oracle" in txt.lower() and "bk_oracle" in txt or "ora_align" in txt or "libbk_oracle" in txt:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_host_packer_round_trip(lib):
    """bk_pack_reads (host code, no device): every base ends up in its 2-bit field or, when it is not a,c,g,t, in the exception list
    with its code; mask and quality bits are dropped; slices packed by different threads join up"""
    import numpy as np
    import biokanga_amd as bk
    rng = np.random.default_rng(1)
    for n in (0, 1, 1000, 70000):
        lens = rng.integers(1, 300, size=n).astype(np.uint32)
        if n:
            lens[0] = 2000
        tot = int(lens.sum())
        bases = rng.integers(0, 4, size=tot).astype(np.uint8)
        if tot:
            if n:
                bases[100:700] = 4                                # a run longer than one entry holds, inside the 2000-base read
            bases[rng.choice(tot, min(tot, 50), replace=False)] = 4
            bases[rng.choice(tot, min(tot, 5), replace=False)] = 6
        raw = bases | (rng.integers(0, 32, size=tot).astype(np.uint8) << 3)
        for explicit in (False, True):
            offs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64) if n else np.zeros(0, np.uint64)
            w, l16, exc = bk.pack_reads(raw, offs if explicit else None, lens)
            assert np.array_equal(l16, lens.astype(np.uint16))
            assert len(w) == int(((lens.astype(np.int64) + 15) // 16).sum())
            # unpack with numpy
            wofs = np.concatenate([[0], np.cumsum((lens.astype(np.int64) + 15) // 16)])[:-1] if n else np.zeros(0, np.int64)
            read_of = np.repeat(np.arange(n), lens)
            pos = np.arange(tot) - np.repeat(offs.astype(np.int64), lens)
            got = (w[wofs[read_of] + pos // 16] >> (30 - 2 * (pos % 16)).astype(np.uint32)) & 3 if tot else np.zeros(0, np.uint32)
            plain = bases < 4
            assert np.array_equal(got[plain], bases[plain])
            e_idx = np.nonzero(~plain)[0]
            # the exceptions come as runs: expand them
            rr = np.repeat(exc["read"], exc["run"].astype(np.int64) + 1)
            pp = np.concatenate([np.arange(int(e["pos"]), int(e["pos"]) + int(e["run"]) + 1) for e in exc]) if len(exc) else np.zeros(0, np.int64)
            cc = np.repeat(exc["code"], exc["run"].astype(np.int64) + 1)
            assert np.array_equal(rr, read_of[e_idx]) and np.array_equal(pp, pos[e_idx]) and np.array_equal(cc, bases[e_idx])
    with pytest.raises(bk.BkError):
        bk.pack_reads(np.zeros(2001, np.uint8), None, np.array([2001], np.uint32))


def test_image_policy_is_one_pure_rule():
    """bk_image_policy - the flags `biokanga align` and bench.py both take from it: lean and growing below BK_POLICY_MIN_READS reads per
    device (or when the count is unknown), every table and the window array from there on; no device needed"""
    import biokanga_amd as bk
    assert bk.POLICY_MIN_READS == 20_000_000
    for n in (0, 1, 1_000_000, bk.POLICY_MIN_READS - 1):
        assert bk.image_policy(n) == bk.CTX_GROW_IMAGE
    for n in (bk.POLICY_MIN_READS, 50_000_000, 125_000_000, 10**12):
        assert bk.image_policy(n) == bk.CTX_WINDOW_ARRAY_EAGER
    # the header's constant is the one the library was built with
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "biokanga_amd.h")).read()
    assert "#define BK_POLICY_MIN_READS 20000000ULL" in hdr
