"""Multi-GPU path on CPU: segment-range sharding of every scheme (pure host logic) and the single
gather collective, exercised with world_size-2 gloo process groups.  The decoder in these CPU tests is
the oracle (the HIP kernels take its place on a GPU box — tests/test_gpu_sharding.py)."""
import os
import socket
import sys

import numpy as np
import pytest

import cases
from libgiddy_amd import formats, sharding
from oracle import oracle


def _decode(scheme, f, out_dtype, **kw):
    n = int(f["length"]) if "length" in f else 32 * int(f["bitmap_length"])
    if n == 0:
        return np.zeros(0, dtype=out_dtype)
    if scheme == "for":
        return oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, int(f["segment_length"]), out_dtype, kw.get("model", "constant"))
    if scheme == "dictionary":
        return oracle.dictionary(np.ascontiguousarray(f["compressed_input"]), f["dictionary_entries"])
    if scheme == "delta":
        return oracle.delta(np.ascontiguousarray(f["compressed_input"]), np.ascontiguousarray(f["anchor_values"]), int(f["segment_length"]))
    if scheme == "dzbv":
        return oracle.dzb_variable(np.ascontiguousarray(f["compressed_data"]), np.ascontiguousarray(f["packed_element_sizes"]),
                                   np.ascontiguousarray(f["position_anchors"]), int(f["segment_length"]), n,
                                   int(f["min_represented_element_size"]), int(f["bits_per_element_size"]), out_dtype)
    if scheme == "rle":
        return oracle.run_length_encoding(np.ascontiguousarray(f["run_data"]), np.ascontiguousarray(f["run_lengths"]),
                                          np.ascontiguousarray(f["position_anchors"]), np.ascontiguousarray(f["intra_run_anchor_offsets"]),
                                          int(f["segment_length"]), n)
    if scheme == "rpe":
        return oracle.run_position_encoding(np.ascontiguousarray(f["run_data"]), np.ascontiguousarray(f["run_start_positions"]),
                                            np.ascontiguousarray(f["position_anchors"]), int(f["segment_length"]), n, kw["relative"])
    if scheme == "model":
        return oracle.model(f["model_coefficients"], n, out_dtype, "linear")
    if scheme == "ib":
        return oracle.incidence_bitmaps(np.ascontiguousarray(f["incidence_bitmaps"]), f["dictionary_entries"], int(f["bitmap_length"]),
                                        int(f["num_bitmaps"]), 2, out_dtype)[0]
    raise KeyError(scheme)


def _columns():
    r = cases.rng(99)
    n = 50000
    out = {}
    f = cases.for_case(n, 1000, np.int32, np.int16, "linear", seed=1)
    out["for"] = (f, oracle.frame_of_reference(f["compressed_input"], f["segment_model_coefficients"], n, 1000, np.int32, "linear"),
                  sharding.frame_of_reference, np.int32, {"model": "linear"})
    f = cases.dictionary_case(n, 256, np.uint32, np.uint8, seed=2)
    out["dictionary"] = (f, oracle.dictionary(f["compressed_input"], f["dictionary_entries"]), sharding.elementwise, np.uint32, {})
    f, truth = cases.delta_case(n, 512, np.uint64, np.uint16, seed=3)
    out["delta"] = (f, truth, sharding.delta, np.uint64, {})
    col, sizes = cases.dzbv_column(n, np.uint64, 1, 3, seed=4)   # 10 sizes per container: shard boundaries must respect lcm
    out["dzbv"] = (formats.discard_zero_bytes_variable(col, 256, 1, 3, np.uint32, 8, element_sizes=sizes), col, sharding.discard_zero_bytes_variable, np.uint64, {})
    col = cases.runs_column(n, 20, np.uint32, seed=5)
    out["rle"] = (formats.run_length_encoding(col, 1024, np.uint16), col, sharding.run_length_encoding, np.uint32, {})
    out["rpe_rel"] = (formats.run_position_encoding(col, 1024, np.uint16, True), col,
                      lambda f, r_, w: sharding.run_position_encoding(f, r_, w, relative=True), np.uint32, {"relative": True})
    out["rpe_abs"] = (formats.run_position_encoding(col, 1024, np.uint32, False), col,
                      lambda f, r_, w: sharding.run_position_encoding(f, r_, w, relative=False), np.uint32, {"relative": False})
    boundary = np.repeat(np.arange(10, dtype=np.uint32) * 3 + 1, [1024, 512, 512, 2048, 1, 1023, 1024, 100, 924, 1000])
    out["rle_boundary"] = (formats.run_length_encoding(boundary, 1024, np.uint16), boundary, sharding.run_length_encoding, np.uint32, {})
    out["rpe_rel_boundary"] = (formats.run_position_encoding(boundary, 1024, np.uint16, True), boundary,
                               lambda f, r_, w: sharding.run_position_encoding(f, r_, w, relative=True), np.uint32, {"relative": True})
    c = np.array([-1000, 37], dtype=np.int32)
    out["model"] = ({"model_coefficients": c, "length": n}, oracle.model(c, n, np.int32, "linear"),
                    lambda f, r_, w: sharding.model(f, r_, w, "linear"), np.int32, {})
    pool = r.choice(1 << 31, size=11, replace=False).astype(np.uint32)
    col = pool[r.integers(0, 11, size=32 * 1000)]
    out["ib"] = (formats.incidence_bitmaps(col, 2), col, sharding.incidence_bitmaps, np.uint32, {})
    return out


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_decode_to_the_column(world):
    for name, (fields, truth, shard_fn, dtype, kw) in _columns().items():
        scheme = name.split("_")[0]
        pieces, covered = [], 0
        for rank in range(world):
            sh = shard_fn(fields, rank, world)
            assert sh.start == covered, (name, rank)
            got = _decode(scheme, sh.fields, dtype, **kw)
            assert len(got) == sh.stop - sh.start
            assert (got == truth[sh.start:sh.stop]).all(), (name, world, rank)
            covered = sh.stop
            pieces.append(got)
        assert covered == len(truth), name
        assert (np.concatenate(pieces) == truth).all()


def _as_tensors(fields):
    """every array of a column as a torch tensor holding the same bits (signed dtypes: torch's unsigned ones lack arithmetic)"""
    import torch
    signed = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}
    return {k: (torch.from_numpy(np.ascontiguousarray(v).view(signed[v.dtype.itemsize]).copy()) if isinstance(v, np.ndarray) and v.dtype.kind in "iu" and k != "model_coefficients" else v)
            for k, v in fields.items()}


def _as_arrays(fields, like):
    return {k: (v.numpy().view(like[k].dtype) if hasattr(v, "numpy") else v) for k, v in fields.items()}


@pytest.mark.parametrize("world", [2, 5])
def test_shards_of_tensors_decode_to_the_column(world):
    """the same cuts on torch tensors (what a device-resident column is): slices are views, anchors are rebased by a
    wrapping tensor subtraction — bit-identical shards to the numpy cuts"""
    for name, (fields, truth, shard_fn, dtype, kw) in _columns().items():
        scheme = name.split("_")[0]
        tensors = _as_tensors(fields)
        covered = 0
        for rank in range(world):
            sh, reference = shard_fn(tensors, rank, world), shard_fn(fields, rank, world)
            assert (sh.start, sh.stop) == (reference.start, reference.stop) and sh.start == covered
            back = _as_arrays(sh.fields, fields)
            for k, v in reference.fields.items():
                if isinstance(v, np.ndarray):
                    assert np.array_equal(np.asarray(back[k]).reshape(-1), v.reshape(-1)), (name, rank, k)
            views = ("compressed_input", "compressed_data", "run_data", "run_lengths") + (("run_start_positions",) if kw.get("relative", True) else ())
            for k in views:   # (absolute run positions are rebased to the shard: the one big array that is rewritten, 1/mean-run of the output)
                if k in sh.fields and sh.stop > sh.start and hasattr(sh.fields[k], "untyped_storage"):
                    assert sh.fields[k].untyped_storage().data_ptr() == tensors[k].untyped_storage().data_ptr(), (name, k, "a copy, not a view")
            got = _decode(scheme, back, dtype, **kw)
            assert (got == truth[sh.start:sh.stop]).all(), (name, world, rank)
            covered = sh.stop
        assert covered == len(truth), name


def test_patch_partition_of_tensors():
    import torch
    n = 100000
    p = cases.patch_case(n, 300, np.uint64, 4, seed=7)
    t = {"indices": torch.from_numpy(p["indices"].view(np.int32).copy()), "data": torch.from_numpy(p["data"].view(np.int64).copy()), "data_length": 300}
    shuffled = np.random.default_rng(1).permutation(300)
    for start, stop in [(0, 25000), (25000, 70000), (70000, n)]:
        want = sharding.patches(p, start, stop)
        got = sharding.patches(t, start, stop)
        assert got["data_length"] == want["data_length"]
        assert (got["indices"].numpy().view(np.uint32) == want["indices"]).all() and (got["data"].numpy().view(np.uint64) == want["data"]).all()
        unordered = sharding.patches({"indices": p["indices"][shuffled], "data": p["data"][shuffled], "data_length": 300}, start, stop)
        assert sorted(unordered["indices"].tolist()) == want["indices"].tolist()


def test_patch_partition():
    n = 100000
    p = cases.patch_case(n, 300, np.uint64, 4, seed=7)
    seen = 0
    for rank in range(4):
        s0, s1 = sharding.segment_range(100, rank, 4)
        q = sharding.patches(p, s0 * 1000, s1 * 1000)
        assert ((q["indices"].astype(np.int64) + s0 * 1000) == p["indices"][seen:seen + q["data_length"]]).all()
        seen += q["data_length"]
    assert seen == 300


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, results):
    import torch
    import torch.distributed as dist
    from libgiddy_amd import distributed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        col = cases.runs_column(40000, 16, np.uint32, seed=11)
        fields = formats.run_length_encoding(col, 1024, np.uint16)
        sh = sharding.run_length_encoding(fields, rank, world)
        local = _decode("rle", sh.fields, np.uint32)
        counts = [None] * world
        dist.all_gather_object(counts, len(local))
        gathered = distributed.gather_decoded(torch.from_numpy(local.view(np.int32).copy()), counts, dst=0)
        if rank == 0:
            ok = bool((gathered.numpy().view(np.uint32) == col).all())
            results.put(("rle", ok))
        # dzb-variable with an uneven split
        colv, sizes = cases.dzbv_column(30000, np.uint64, 1, 2, seed=12)
        f = formats.discard_zero_bytes_variable(colv, 1024, 1, 2, np.uint32, 8, element_sizes=sizes)
        sh = sharding.discard_zero_bytes_variable(f, rank, world)
        local = _decode("dzbv", sh.fields, np.uint64)
        dist.all_gather_object(counts, len(local))
        gathered = distributed.gather_decoded(torch.from_numpy(local.view(np.int64).copy()), counts, dst=0)
        if rank == 0:
            results.put(("dzbv", bool((gathered.numpy().view(np.uint64) == colv).all())))
        # ... and the same column gathered COMPRESSED: the peers' shards arrive in pieces, rank 0 decodes them into place
        tensors_v = _as_tensors(f)
        shv = sharding.discard_zero_bytes_variable(tensors_v, rank, world)
        whole = torch.zeros(len(colv), dtype=torch.int64) if rank == 0 else None
        def decode_into(fields, out_view):
            out_view.copy_(torch.from_numpy(_decode("dzbv", _as_arrays(fields, f), np.uint64).view(np.int64).copy()))
        moved = distributed.gather_compressed(shv.fields, sharding.discard_zero_bytes_variable, decode_into, whole, counts, dst=0, pieces=3)
        if rank == 0:
            results.put(("dzbv_compressed", bool((whole.numpy().view(np.uint64) == colv).all()) and 0 < moved[0] < moved[1]))
        tensors_r = _as_tensors(fields)
        shr = sharding.run_length_encoding(tensors_r, rank, world)
        counts_r = [None] * world
        dist.all_gather_object(counts_r, shr.stop - shr.start)
        whole = torch.zeros(len(col), dtype=torch.int32) if rank == 0 else None
        def decode_runs(fs, out_view):
            out_view.copy_(torch.from_numpy(_decode("rle", _as_arrays(fs, fields), np.uint32).view(np.int32).copy()))
        distributed.gather_compressed(shr.fields, sharding.run_length_encoding, decode_runs, whole, counts_r, dst=0, pieces=4)
        if rank == 0:
            results.put(("rle_compressed", bool((whole.numpy().view(np.uint32) == col).all())))
        # bench.py's gather leg (the timing harness around the same collective), on CPU tensors, uneven shards
        import bench
        column = torch.arange(9000, dtype=torch.int32) * 7 + 3
        cut = [0, 5000, 9000] if world == 2 else [9000 * r // world for r in range(world + 1)]
        local_t = column[cut[rank]:cut[rank + 1]].clone()
        report = bench.time_gather(local_t, [cut[r + 1] - cut[r] for r in range(world)], rank, world, torch.device("cpu"), dist.barrier,
                                   repetitions=1, expected=column if rank == 0 else None)
        if rank == 0:
            results.put(("bench_gather", report["own_shard_in_place"] and report["gathered_equals_the_column"] and report["links"] == world - 1))
        # bench.py's strong-scaling check: shards of tensor columns (cut as on a device), position-weighted checksums summed over the ranks
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        import workloads
        tensors = _as_tensors(fields)
        sh = sharding.run_length_encoding(tensors, rank, world)
        decoded = torch.from_numpy(_decode("rle", _as_arrays(sh.fields, fields), np.uint32).view(np.int32).copy())
        summed = bench.all_reduce_u64_sum(workloads.checksum(decoded, first_index=sh.start), world, torch.device("cpu"), "gloo")
        if rank == 0:
            results.put(("strong_checksum", summed == workloads.checksum(torch.from_numpy(col.view(np.int32).copy()))))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_gather_with_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    results = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(results.get(timeout=10) for _ in range(6))
    assert got == {"rle": True, "dzbv": True, "dzbv_compressed": True, "rle_compressed": True, "bench_gather": True, "strong_checksum": True}
