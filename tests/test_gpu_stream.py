"""GPU tests of the overlapped pipeline (bk_stream_*): batches submitted from host buffers must give exactly the
records the blocking batch call and the CPU oracle give, whatever the batch size, buffer kind (pinned / pageable),
offset form (explicit / back-to-back) or pipeline depth; list modes and paired ends travel with their batch."""
import os

import numpy as np
import pytest

import helpers
from test_gpu_parity import FIELDS, assert_hits_equal, load_fixture, _bk

pytestmark = pytest.mark.gpu


def _contiguous(bases, offs, lens):
    """reads copied back to back (what offs=None means)"""
    out = np.concatenate([bases[int(o):int(o) + int(l)] for o, l in zip(offs, lens)]) if len(lens) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(out, dtype=np.uint8)


@pytest.mark.parametrize("fixture,tag", [("basic", "s3"), ("repeat", "s3"), ("lengths", "s3L")])
@pytest.mark.parametrize("batch,depth,pinned,explicit_offs", [(257, 3, True, True), (1000, 2, False, False), (64, 4, True, False),
                                                               (100000, 1, False, True), (257, 3, True, "packed"), (1000, 2, False, "packed"),
                                                               (100000, 1, True, "packed")])
def test_stream_equals_blocking_call_and_oracle(golden_tmp, fixture, tag, batch, depth, pinned, explicit_offs):
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, tag)
    offs, lens = offs[keep], lens[keep]
    n = len(lens)
    cb = _contiguous(bases, offs, lens)
    coffs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    mk = (lambda m, dt: bk.host_array(m, dt)) if pinned else (lambda m, dt: np.zeros(m, dtype=dt))
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        ref = al.align(bases, offs, lens)
        out = mk(n, bk.HIT_DTYPE)
        out[:] = np.zeros(1, bk.HIT_DTYPE)[0]
        with bk.Stream(al, max_batch_reads=min(batch, n), max_batch_bases=int(lens.astype(np.int64).max()) * min(batch, n), depth=depth) as st:
            tickets = []
            for lo in range(0, n, batch):
                hi = min(n, lo + batch)
                b0, b1 = int(coffs[lo]), int(coffs[hi - 1]) + int(lens[hi - 1])
                hb = mk(b1 - b0, np.uint8); hb[:] = cb[b0:b1]
                hl = mk(hi - lo, np.uint32); hl[:] = lens[lo:hi]
                ho = None
                if explicit_offs == "packed":
                    tickets.append(st.submit_packed(*bk.pack_reads(hb, None, hl, pinned=pinned), out[lo:hi]))
                    continue
                if explicit_offs:
                    ho = mk(hi - lo, np.uint64); ho[:] = coffs[lo:hi] - np.uint64(b0)
                tickets.append(st.submit(hb, ho, hl, out[lo:hi]))
            for t in tickets:
                st.wait(t)
            stats = st.stats()
        assert stats["reads"] == n and stats["batches"] == len(tickets)
        assert stats["bytes_d2h"] == 20 * n
        assert stats["seconds_first_submit_to_last_result"] > 0
    assert_hits_equal(out, ref, [names[i] for i in keep])
    sfx = helpers.OracleSfx(os.path.join(d, "genome.sfx"))
    exp, _ = sfx.align(bases, offs, lens, helpers.make_params(max_subs=3))
    sfx.close()
    assert_hits_equal(out, exp, [names[i] for i in keep])


def test_stream_rejects_reads_outside_the_batch(golden_tmp):
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "basic", "s3")
    with bk.Aligner(os.path.join(d, "genome.sfx"), bk.AlignParams(max_subs=3)) as al:
        with bk.Stream(al, 16, 4096, depth=2) as st:
            hb = np.zeros(1000, np.uint8)
            hl = np.full(4, 100, np.uint32)
            ho = np.array([0, 100, 950, 300], dtype=np.uint64)          # read 2 runs past the 1000 bases handed over
            out = np.zeros(4, bk.HIT_DTYPE)
            t = st.submit(hb, ho, hl, out)
            with pytest.raises(bk.BkError) as e:
                st.wait(t)
            assert e.value.rc == -100
            # an offset near 2^64 whose end wraps to a small number is refused as well (checked per read, not by a sum)
            howrap = np.array([0, 100, 0xFFFFFFFFFFFFFFF0, 300], dtype=np.uint64)
            t = st.submit(hb, howrap, hl, out)
            with pytest.raises(bk.BkError) as e:
                st.wait(t)
            assert e.value.rc == -100
            # the stream stays usable
            ho2 = np.array([0, 100, 900, 300], dtype=np.uint64)
            t = st.submit(hb, ho2, hl, out)
            st.wait(t)
            with pytest.raises(bk.BkError):                             # more reads than the stream was sized for
                st.submit(np.zeros(1700, np.uint8), None, np.full(17, 100, np.uint32), np.zeros(17, bk.HIT_DTYPE))


def test_stream_multi_loci_lists_travel_with_their_batch(golden_tmp):
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "repeat", "s3")
    offs, lens = offs[keep], lens[keep]
    n = len(lens)
    p = bk.AlignParams(max_subs=3, max_ml=5)
    with bk.Aligner(os.path.join(d, "genome.sfx"), p) as al:
        ref = al.align(bases, offs, lens)
        ro, rl = al.batch_loci(n)
        cb = _contiguous(bases, offs, lens)
        coffs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
        out = np.zeros(n, bk.HIT_DTYPE)
        B = 301
        got_offs, got_loci = [np.zeros(1, np.uint64)], []
        with bk.Stream(al, B, int(lens.max()) * B, depth=3) as st:
            tk = []
            for lo in range(0, n, B):
                hi = min(n, lo + B)
                b0, b1 = int(coffs[lo]), int(coffs[hi - 1]) + int(lens[hi - 1])
                tk.append((st.submit(cb[b0:b1].copy(), None, lens[lo:hi].copy(), out[lo:hi]), hi - lo))
            for t, m in tk:
                st.wait(t)
                o, l = st.batch_loci(t, m)
                got_offs.append(o[1:] + got_offs[-1][-1])
                got_loci.append(l)
                st.release(t)
    assert_hits_equal(out, ref)
    assert np.array_equal(np.concatenate(got_offs), ro)
    assert np.array_equal(np.concatenate(got_loci), rl)


@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("fixture", ["pe", "pe150"])
def test_stream_paired_end_association_on_resident_buffers(golden_tmp, tmp_path, fixture, packed):
    from test_oracle_pe import pe_cfg, pe_inputs, check_pe_hits_against_sam
    bk = _bk()
    cfg = pe_cfg(fixture, "U3")
    names, bases, offs, lens = pe_inputs(tmp_path, fixture)
    cb = _contiguous(bases, offs, lens)
    coffs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    n = len(lens)
    pe = bk.PEParams(cfg["pe"], cfg["d"], cfg["D"], cfg.get("E", False))
    with bk.Aligner(os.path.join(golden_tmp["basic"], "genome.sfx"), bk.AlignParams(max_subs=cfg["s"])) as al:
        ref = al.pair(bases, offs, lens, al.align(bases, offs, lens), pe)
        out = np.zeros(n, bk.HIT_DTYPE)
        B = 2 * 77
        with bk.Stream(al, B, int(lens.max()) * B, depth=2, pe=pe) as st:
            tk = []
            for lo in range(0, n, B):
                hi = min(n, lo + B)
                b0, b1 = int(coffs[lo]), int(coffs[hi - 1]) + int(lens[hi - 1])
                if packed:
                    tk.append(st.submit_packed(*bk.pack_reads(cb[b0:b1], None, lens[lo:hi]), out[lo:hi]))
                else:
                    tk.append(st.submit(cb[b0:b1].copy(), None, lens[lo:hi].copy(), out[lo:hi]))
            for t in tk:
                st.wait(t)
            with pytest.raises(bk.BkError):                 # half a pair
                st.submit(cb[:int(lens[0])].copy(), None, lens[:1].copy(), out[:1])
    for f in FIELDS + ["flags"]:
        assert np.array_equal(out[f], ref[f]), f
    check_pe_hits_against_sam(names, out, "U3", ["chrA", "chrB"], fixture)


@pytest.mark.parametrize("packed", [False, True])
def test_stream_paired_ends_with_chimeric_trimming(golden_tmp, tmp_path, packed):
    """a pipeline with pe on a context that trims chimeric reads: the association sees each batch's bk_seg2 records and the batch's
    records come back with the trims of the recovered partners"""
    from test_oracle_pe import PECHIM_RUNS, pe_inputs, check_pechim_against_sam
    bk = _bk()
    cfg = PECHIM_RUNS["U3c50"]
    names, bases, offs, lens = pe_inputs(tmp_path, "pechim")
    cb = _contiguous(bases, offs, lens)
    coffs = np.concatenate([[0], np.cumsum(lens[:-1], dtype=np.uint64)]).astype(np.uint64)
    n = len(lens)
    pe = bk.PEParams(cfg["pe"], cfg["d"], cfg["D"], False)
    with bk.Aligner(os.path.join(golden_tmp["chimeric"], "genome.sfx"), bk.AlignParams(max_subs=cfg["s"], min_chimeric_len=cfg["c"])) as al:
        se = al.align(bases, offs, lens)
        ref, ref_seg = al.pair(bases, offs, lens, se, pe, seg2=al.batch_seg2())
        out = np.zeros(n, bk.HIT_DTYPE)
        segs = []
        B = 2 * 90
        with bk.Stream(al, B, int(lens.max()) * B, depth=2, pe=pe) as st:
            tk = []
            for lo in range(0, n, B):
                hi = min(n, lo + B)
                b0, b1 = int(coffs[lo]), int(coffs[hi - 1]) + int(lens[hi - 1])
                if packed:
                    tk.append(st.submit_packed(*bk.pack_reads(cb[b0:b1], None, lens[lo:hi]), out[lo:hi]))
                else:
                    tk.append(st.submit(cb[b0:b1].copy(), None, lens[lo:hi].copy(), out[lo:hi]))
            for t in tk:
                st.wait(t)
                segs.append(st.batch_seg2(t))
                st.release(t)
    seg = np.concatenate(segs)
    for f in FIELDS + ["flags"]:
        assert np.array_equal(out[f], ref[f]), f
    assert np.array_equal(seg, ref_seg)
    check_pechim_against_sam(names, out, seg, "U3c50")


@pytest.mark.parametrize("fixture,pe_tag", [("basic", None), ("pe", "U3")])
def test_stream_submit_device_is_asynchronous_and_ordered_behind_the_producer(golden_tmp, tmp_path, fixture, pe_tag):
    """bk_stream_submit_device: buffers in HBM, the call returns before the batch has been aligned and the batch runs after the work the
    caller had enqueued on its stream - here the copies that FILL the read buffers, issued on a side stream right before the submit
    (submitted with the buffers still holding rubbish, the results must be those of the real reads)"""
    import torch
    bk = _bk()
    if pe_tag:
        from test_oracle_pe import pe_cfg, pe_inputs
        cfg = pe_cfg(fixture, pe_tag)
        names, bases, offs, lens = pe_inputs(tmp_path, fixture)
        sfx, kw = os.path.join(golden_tmp["basic"], "genome.sfx"), dict(max_subs=cfg["s"])
        pe = bk.PEParams(cfg["pe"], cfg["d"], cfg["D"], cfg.get("E", False))
    else:
        d, names, bases, offs, lens, keep = load_fixture(golden_tmp, fixture, "s3")
        offs, lens = offs[keep], lens[keep]
        sfx, kw, pe = os.path.join(d, "genome.sfx"), dict(max_subs=3), None
    n = len(lens)
    dev = torch.device("cuda", 0)
    with bk.Aligner(sfx, bk.AlignParams(**kw)) as al:
        ref = al.align(bases, offs, lens)
        if pe:
            ref = al.pair(bases, offs, lens, ref, pe)
        h_b = torch.from_numpy(np.ascontiguousarray(bases)).pin_memory()
        h_o = torch.from_numpy(offs.astype(np.int64)).pin_memory()
        h_l = torch.from_numpy(lens.astype(np.int32)).pin_memory()
        d_b = torch.full((len(bases),), 9, dtype=torch.uint8, device=dev)
        d_o = torch.zeros(n, dtype=torch.int64, device=dev)
        d_l = torch.full((n,), 17, dtype=torch.int32, device=dev)
        outs = [torch.zeros(n * bk.HIT_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(3)]
        side = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        with bk.Stream(al, max(n, 2), int(lens.sum()) + 16, depth=3, pe=pe) as st:
            with torch.cuda.stream(side):
                d_b.copy_(h_b, non_blocking=True)
                d_o.copy_(h_o, non_blocking=True)
                d_l.copy_(h_l, non_blocking=True)
            tk = [st.submit_device(d_b.data_ptr(), d_o.data_ptr(), d_l.data_ptr(), n, o.data_ptr(), side.cuda_stream) for o in outs]
            h_mixed = np.zeros(n, bk.HIT_DTYPE)                            # a host-form batch between device-form ones keeps its place
            cb = _contiguous(bases, offs, lens)
            tk_h = st.submit(cb.copy(), None, lens.copy(), h_mixed)
            for t in tk:
                st.wait(t)
            st.wait(tk_h)
            with pytest.raises(bk.BkError):
                st.submit_device(0, d_o.data_ptr(), d_l.data_ptr(), n, outs[0].data_ptr(), None)
    for o in outs:
        got = o.cpu().numpy().view(bk.HIT_DTYPE)
        for f in FIELDS + ["flags"]:
            assert np.array_equal(got[f], ref[f]), f
    for f in FIELDS + ["flags"]:
        assert np.array_equal(h_mixed[f], ref[f]), f


def test_seq_counts_allreduce_over_contexts(golden_tmp):
    """the exchange step: two contexts, each fed half of the reads, reduce to the counts of one context fed all of them"""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "basic", "s3")
    offs, lens = offs[keep], lens[keep]
    sfx = os.path.join(d, "genome.sfx")
    with bk.Aligner(sfx, bk.AlignParams(max_subs=3)) as a0, bk.Aligner(sfx, bk.AlignParams(max_subs=3)) as a1, \
            bk.Aligner(sfx, bk.AlignParams(max_subs=3)) as whole:
        hits = whole.align(bases, offs, lens)
        exp = whole.seq_counts()
        a0.align(bases, offs[0::2], lens[0::2])
        a1.align(bases, offs[1::2], lens[1::2])
        got = bk.seq_counts_allreduce([a0, a1], reset=True)
        assert np.array_equal(got, exp)
        assert got.sum() == int((hits["nar"] == 1).sum())
        assert a0.seq_counts().sum() == 0 and a1.seq_counts().sum() == 0          # reset


def test_exchange_step_through_rccl_on_one_device(golden_tmp):
    """the library's own RCCL binding on hardware: dlopen of librccl, ncclCommInitAll (a communicator of one rank), the grouped
    ncclAllReduce on the context's stream - forced for contexts that share the one GPU ("force_rccl"), where the reduction would
    otherwise be a device-side add.  Same counts as the plain path, twice (communicators are kept)."""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "basic", "s3")
    offs, lens = offs[keep], lens[keep]
    sfx = os.path.join(d, "genome.sfx")
    with bk.Aligner(sfx, bk.AlignParams(max_subs=3)) as a0, bk.Aligner(sfx, bk.AlignParams(max_subs=3)) as a1:
        hits = a0.align(bases, offs, lens)
        exp = a0.seq_counts(reset=True)
        assert a0.tune("rccl_allreduces", 0) == 0
        a0.tune("force_rccl", 1)
        a0.align(bases, offs, lens)
        got = bk.seq_counts_allreduce([a0], reset=True)
        assert np.array_equal(got, exp) and got.sum() == int((hits["nar"] == 1).sum())
        assert a0.tune("rccl_allreduces", 0) == 1 and a0.tune("rccl_ranks", 0) == 1
        # two contexts on the device: added up on the device first, then the one-rank all-reduce
        a0.align(bases, offs[0::2], lens[0::2])
        a1.align(bases, offs[1::2], lens[1::2])
        assert np.array_equal(bk.seq_counts_allreduce([a0, a1], reset=True), exp)
        assert a0.tune("rccl_allreduces", 0) == 2 and a1.tune("rccl_allreduces", 0) == 1
        a0.tune("force_rccl", 0)
        a0.align(bases, offs, lens)
        assert np.array_equal(bk.seq_counts_allreduce([a0], reset=True), exp)
        assert a0.tune("rccl_allreduces", 0) == 2


def test_cloned_context_and_distinct_devices(golden_tmp):
    """bk_ctx_clone copies the finished index image device to device; a clone aligns like its source.  With two GPUs visible the
    clone lives on the second one and the exchange step takes the RCCL branch (ncclAllReduce between distinct devices)."""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "basic", "s3")
    offs, lens = offs[keep], lens[keep]
    sfx = os.path.join(d, "genome.sfx")
    second = 1 if bk.device_count() >= 2 else 0
    with bk.Aligner(sfx, bk.AlignParams(max_subs=3)) as a0:
        ref = a0.align(bases, offs, lens)
        exp = a0.seq_counts(reset=True)
        with bk.Aligner(clone_of=a0, device=second) as a1:
            got = a1.align(bases, offs, lens)
            assert_hits_equal(got, ref)
            assert np.array_equal(a1.seq_counts(reset=True), exp)
            a0.align(bases, offs[0::2], lens[0::2])
            a1.align(bases, offs[1::2], lens[1::2])
            assert np.array_equal(bk.seq_counts_allreduce([a0, a1], reset=True), exp)
            # communicators are kept: a second reduction over the same devices
            a0.align(bases, offs[1::2], lens[1::2])
            a1.align(bases, offs[0::2], lens[0::2])
            assert np.array_equal(bk.seq_counts_allreduce([a1, a0], reset=True), exp)


@pytest.mark.skipif("__import__('biokanga_amd').device_count() < 2", reason="needs two GPUs: the RCCL branch between distinct devices")
def test_seq_counts_allreduce_over_distinct_devices(golden_tmp):
    """three contexts on two devices: the two sharing a GPU are added up there, RCCL sums across the GPUs"""
    bk = _bk()
    d, names, bases, offs, lens, keep = load_fixture(golden_tmp, "basic", "s3")
    offs, lens = offs[keep], lens[keep]
    sfx = os.path.join(d, "genome.sfx")
    with bk.Aligner(sfx, bk.AlignParams(max_subs=3), device=0) as a0, bk.Aligner(sfx, bk.AlignParams(max_subs=3), device=1) as a1, \
            bk.Aligner(clone_of=a1, device=1) as a2, bk.Aligner(sfx, bk.AlignParams(max_subs=3)) as whole:
        hits = whole.align(bases, offs, lens)
        exp = whole.seq_counts()
        a0.align(bases, offs[0::3], lens[0::3])
        a1.align(bases, offs[1::3], lens[1::3])
        a2.align(bases, offs[2::3], lens[2::3])
        got = bk.seq_counts_allreduce([a0, a1, a2], reset=True)
        assert np.array_equal(got, exp)
        assert got.sum() == int((hits["nar"] == 1).sum())
