"""Network executor (csrc/net.hip, openscene_amd/executor.py), checked WITHOUT a GPU:
  * the stage program compiled from each MinkUNet variant covers the module tree (one stage per convolution, one
    producer per buffer, ME.cat buffers fully written by their two producers) and the library accepts it;
  * the C side plans the same kernels the per-module path picks (functional.py), for the S100k level sizes;
  * DRY RUN: with a null HIP runtime linked in (tools/dryrun), one training step through the executor issues exactly
    the convolution and batch-norm launches (kernel instance + grid) the per-module path issues."""
import collections
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S100K = [100999, 47618, 12868, 3052, 700]


@pytest.mark.parametrize("arch", ["MinkUNet14A", "MinkUNet18A", "MinkUNet18D", "MinkUNet34A", "MinkUNet34C"])
def test_program_covers_the_module_tree(arch):
    import __graft_entry__ as ge
    ge.build()
    from openscene_amd import executor as E
    from openscene_amd import minkowski as ME
    from openscene_amd.mink_unet import mink_unet
    model = mink_unet(3, 768, 3, arch)
    ex = E.for_model(model)
    assert ex is not None
    p = ex.program
    convs = [m for m in model.modules() if isinstance(m, ME.MinkowskiConvolutionBase)]
    norms = [m for m in model.modules() if isinstance(m, ME.MinkowskiBatchNorm)]
    assert len(p.ops) == len(convs) and set(map(id, p.convs)) == set(map(id, convs))
    assert len(p.bns) == len(norms) and set(map(id, p.bns)) == set(map(id, norms))
    assert {id(q) for q in p.params} == {id(q) for q in model.parameters()}
    producers = collections.Counter(o["dst"] for o in p.ops if o["dst"] >= 0)
    assert all(v == 1 for v in producers.values())
    # every ME.cat buffer (never a dst) is written column-complete by exactly two second stores: [up-branch | skip]
    cats = set(range(len(p.bufs))) - set(producers)
    assert len(cats) == 4
    for c in cats:
        parts = sorted((o["copy_col"], o["cout"], o["transposed"]) for o in p.ops if o["copy_buf"] == c)
        assert len(parts) == 2 and parts[0][0] == 0 and parts[0][2] == 1                 # the transposed conv's half first
        assert parts[0][1] == parts[1][0] and parts[1][0] + parts[1][1] == p.bufs[c][1]
        readers = [o for o in p.ops if o["src"] == c]
        assert len(readers) == 2 and {o["K"] for o in readers} == {27, 1}                  # conv1 and the 1x1 shortcut
    # every buffer that is read has a producer or is a cat buffer; residuals are row-aligned with their stage
    for o in p.ops:
        for b in (o["src"], o["res"]):
            assert b == -1 or b in producers or b in cats
    assert p.ops[-1]["dst"] == -1 and p.ops[-1]["K"] == 1 and p.ops[0]["src"] == -1 and p.ops[0]["need_dgrad"] == 0
    # the library accepts the program and plans every stage
    ks = ex.kernels(S100K, training=True)
    assert len(ks) == len(p.ops) and all(k[1] != "none" and k[3] != "none" for k in ks)
    assert ks[0][1:] == ("stem", "none", "wgrad_stem")
    assert int(ex._plan.fwd_arena_bytes) > 4 * S100K[0] * 96 * 10


def test_planned_kernels_are_the_per_module_choice():
    """functional.SparseConvFunction's dispatch, restated: the stem kernel for 3 -> 32; the weight-stationary kernel in
    direct mode for every launch that writes the fine side of a 2^3 stride-2 map; tile-list kernels from TL_FWD_MIN_ROWS
    table rows on (TL_MID_MIN_ROWS for >= 96 channels); the weight-stationary kernel with partial rows for launches writing
    at most WS_MAX_ROWS rows; the split-bf16 output-stationary kernel for the rest; pair-array weight gradient on every
    3^3 / 2^3 map and (identity map) for the 1x1 shortcuts up to 128 channels, the table weight gradient for the stem
    and the 96 -> 768 head."""
    from openscene_amd import executor as E
    from openscene_amd import functional as F_
    from openscene_amd.mink_unet import mink_unet
    ex = E.for_model(mink_unet(3, 768, 3, "MinkUNet18A"))
    ks = ex.kernels(S100K, training=True)

    def want(o, c_src, c_dst, n_src, n_dst, dst_fine):
        if o["K"] == 1:
            return "dense"
        ws = F_.ws_kernel(o["K"], c_src, c_dst, n_src, n_dst, bool(o["fine_unique"]), dst_fine)
        if ws == "ws_direct":
            return ws
        narrow = c_src in (32, 64) and c_dst in (32, 64)
        if F_.tl_rows_ok(n_dst, o["cin"], o["cout"]) and not (narrow and min(c_src, c_dst) == 32):
            return "tl"
        if narrow:
            return "rg"                             # round 6: the narrow layers, before the partial-row weight-stationary kernel
        return ws or "x6"
    for (i, kf, kd, kw), o in zip(ks, ex.program.ops):
        n_in, n_out = S100K[o["lvl_in"]], S100K[o["lvl_out"]]
        if o["K"] == 125:
            continue
        assert o["fine_unique"] == int(o["K"] == 8)
        want_f = want(o, o["cin"], o["cout"], n_in, n_out, bool(o["transposed"]))
        want_d = want(o, o["cout"], o["cin"], n_out, n_in, not o["transposed"])
        assert (kf, kd) == (want_f, want_d), (i, o, kf, kd)
        assert kw == ("wgrad_tl" if (o["K"] > 1 or max(o["cin"], o["cout"]) <= 256) else "wgrad")
    # level 0 (101 k rows): every 3^3 conv; levels 1 and 2 (48 k / 13 k rows): the decoder's >= 96-channel 3^3 convs
    assert sum(k[1] == "tl" for k in ks) == 12 and sum(k[2] == "tl" for k in ks) >= 10
    # the four transposed convs forward, the four strided convs backward: direct; the 3^3 convs of the two deepest levels
    # (3 k and 730 rows) and the 2^3 launches that write them: partial rows
    assert sum(k[1] == "ws_direct" for k in ks) == 4 and sum(k[2] == "ws_direct" for k in ks) == 4
    # (round 6: the 64 -> 64 2^3 launch that writes the 3 k-row level went from the weight-stationary to the register-gather kernel)
    assert sum(k[1] == "ws" for k in ks) == 12 + 1 and sum(k[2] == "ws" for k in ks) == 12 + 2
    # register gather: the four 32 -> 32 layers of level 1, the four 32 / 64-channel layers of level 2, the three 2^3 strided convs
    # between them (forward); the same 3^3 layers backward (the strided convs' input gradients are direct weight-stationary launches)
    assert sum(k[1] == "rg" for k in ks) == 4 + 4 + 3 and sum(k[2] == "rg" for k in ks) == 4 + 4
    assert not any(k[1] == "x6" or k[2] == "x6" for k in ks), "MinkUNet18A no longer launches the first-generation kernel"
    assert not any(k[1] == "x6" and o["K"] > 1 and S100K[o["lvl_out"]] <= 4096 for k, o in zip(ks, ex.program.ops))


def test_executor_is_not_used_outside_its_configuration(monkeypatch):
    from openscene_amd import executor as E
    from openscene_amd import functional as F_
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor

    class FakeX:
        tensor_stride = 1
        F = torch.ones(5, 3)
    ex = E.for_model(mink_unet(3, 20, 3, "MinkUNet14A"))
    assert not ex.usable(FakeX())                       # host tensor: the executor (like every op) needs device memory
    monkeypatch.setattr(E, "_DRY_RUN", True)
    assert ex.usable(FakeX())
    monkeypatch.setattr(F_, "CONV_MODE", "fp32")
    assert not ex.usable(FakeX())
    monkeypatch.setattr(F_, "CONV_MODE", "tl")
    monkeypatch.setattr(E, "ENABLED", False)
    assert not ex.usable(FakeX())


@pytest.mark.parametrize("arch", ["MinkUNet18A", "MinkUNet34C"])
def test_dry_run_executor_issues_the_launches_of_the_module_path(arch, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tools", "dryrun"))
    import dry_step
    lib = dry_step.install(dry_step.build_dry_lib(), monkeypatch.setattr)
    logs, _step, sizes = dry_step.step_logs(lib, arch, points=12000, setattr_=monkeypatch.setattr)
    assert sizes[0] > 6000
    a, b = dry_step.conv_launches(logs["modules"]), dry_step.conv_launches(logs["executor"])
    assert len(a) > 100 and collections.Counter(a) == collections.Counter(b)
    assert any("spconv_tl_kernel" in l for l in b) and any("wgrad_tl_kernel" in l for l in b) and any("stem_fwd" in l for l in b)
    assert collections.Counter(dry_step.bn_launches(logs["modules"])) == collections.Counter(dry_step.bn_launches(logs["executor"]))
    # same order up to the first BasicBlock shortcut (the executor issues a block's 1x1 shortcut before its second conv)
    assert a[:5] == b[:5]
    # what the executor does NOT launch: the elementwise add / cat kernels of the module path
    assert "cat2_kernel" in logs["modules"] and "cat2_kernel" not in logs["executor"]


def test_dry_run_training_mode_forward_without_grad_gets_the_workspace_the_library_plans(monkeypatch):
    """model.train() under torch.no_grad() as the FIRST call of a process (empty workspace pool): the executor used to ask
    osn_net_plan_query for the inference plan (no backward follows) while osn_net_forward lays out for run.training = 1, whose
    workspace also covers the backward kernels -- the run's size check failed unless an earlier call had grown the pooled
    buffer (found on the GPU by tests/test_golden_unet.py running second in the suite).  The real host code of net.hip runs
    here (null HIP runtime), size checks included."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "dryrun"))
    import dry_step
    dry_step.install(dry_step.build_dry_lib(), monkeypatch.setattr)
    from openscene_amd import executor, ops, synthetic as syn
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import CoordinateManager, SparseTensor
    monkeypatch.setattr(executor, "ENABLED", True)
    torch.manual_seed(0)
    model = mink_unet(3, 16, 3, "MinkUNet18A")
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(0, n_pts=12000), 0.04), 0)
    coords = torch.from_numpy(syn.batch_coords([vox]))
    feats = torch.ones(coords.shape[0], 3)
    for train in (False, True):                              # the order of the golden test: eval first, then train, both without grad
        monkeypatch.setattr(ops, "_ws_pool", {})             # a fresh process: nothing pooled yet
        model.train(train)
        with torch.no_grad():
            out = model(SparseTensor(feats, coordinate_manager=CoordinateManager(coords)))
        assert out.shape == (coords.shape[0], 16)
    ex = executor.for_model(model)
    rows = [12, 6, 3, 2, 1]
    ex._plan_query(ops._prep(None), [r * 1000 for r in rows], False)
    inference_ws = int(ex._plan.ws_bytes)
    ex._plan_query(ops._prep(None), [r * 1000 for r in rows], True)
    assert int(ex._plan.ws_bytes) >= inference_ws


@pytest.mark.parametrize("arch", ["MinkUNet18A", "MinkUNet34C"])
def test_dry_run_every_call_gets_by_with_exactly_the_workspace_it_asked_for(arch, monkeypatch):
    """The workspace pool hands out its largest buffer so far, which hides a request that is too small until a process happens
    to make that call first (see the test above).  Here every request is served with EXACTLY the bytes asked for, and the real
    entry points (null HIP runtime) check their sizes: per-module path and executor, training step, evaluation and
    training-mode forward without grad, one and two scenes."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "dryrun"))
    import dry_step
    dry_step.install(dry_step.build_dry_lib(), monkeypatch.setattr)
    from openscene_amd import executor, functional as F_, ops, synthetic as syn
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import CoordinateManager, SparseTensor
    asked = []

    def exact(nbytes, dev):
        asked.append(int(nbytes))
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8)
    monkeypatch.setattr(ops, "_ws", exact)
    monkeypatch.setattr(F_, "TL_FWD_MIN_ROWS", 6000)         # small scenes: still send level 0 through the tile-list kernels
    torch.manual_seed(0)
    model = mink_unet(3, 32, 3, arch)
    for n_scenes in (1, 2):
        vox = [syn.shuffled(syn.grid_voxels(syn.room_points(b, n_pts=12000), 0.04), b) for b in range(n_scenes)]
        coords = torch.from_numpy(syn.batch_coords(vox))
        feats = torch.ones(coords.shape[0], 3)
        for use_executor in (False, True):
            monkeypatch.setattr(executor, "ENABLED", use_executor)
            model.train()
            out = model(SparseTensor(feats, coordinate_manager=CoordinateManager(coords)))
            model.zero_grad(set_to_none=True)
            out.sum().backward()
            for train in (False, True):
                model.train(train)
                with torch.no_grad():
                    out = model(SparseTensor(feats, coordinate_manager=CoordinateManager(coords)))
                assert out.shape == (coords.shape[0], 32)
    assert len(asked) > 100 and max(asked) > 1 << 20


def test_dry_run_stream_queueing_of_a_training_step(monkeypatch):
    """What the executor queues where (net.hip; measured in profiles/r04_s15_to_s19_stream_queueing_ab.txt), read from the null
    runtime's log of one training step (kernel, stream, event records / waits):
      * forward: every shortcut stage (1x1 conv + batch norm) runs on the side stream, and the event it waits for was recorded
        on the main stream IN FRONT of the block's conv1 -- at least one main-stream convolution is queued between the record
        and the wait;  the main stream waits once per shortcut stage (the residual's join) and for nothing else;
      * backward: the main stream waits exactly once (the pass's final join): no join behind the side stream's backlog;
        every pair-array weight gradient runs on the side stream, the stem's weight gradient on the main stream, and the last
        batched reduction of the pair-array gradients is queued before the stem's weight gradient."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "dryrun"))
    import dry_step
    lib = dry_step.install(dry_step.build_dry_lib(), monkeypatch.setattr)
    logs, _step, _sizes = dry_step.step_logs(lib, "MinkUNet18A", points=12000, setattr_=monkeypatch.setattr)
    recs = []                                                # (kind, text, stream)
    for line in logs["executor"].split("\n"):
        if line.startswith("S ") and recs and recs[-1][0] == "K":
            recs[-1] = ("K", recs[-1][1], line[2:])
        elif line.startswith("K "):
            recs.append(("K", line[2:], None))
        elif line.startswith(("EVENT ", "WAIT ")):
            kind, ev, st = line.split()
            recs.append((kind, ev, st[1:]))
    main, side = "0", "51de"
    first_bwd = next(i for i, r in enumerate(recs) if r[0] == "K" and ("col_reduce_kernel<1>" in r[1] or "wgrad" in r[1]))
    fwd, bwd = recs[:first_bwd], recs[first_bwd:]
    # ---- forward
    shortcut = [i for i, r in enumerate(fwd) if r[0] == "K" and "dense_kernel" in r[1] and r[2] == side]
    assert len(shortcut) == 7                                # MinkUNet18A: blocks 2 - 8 change their width
    for i in shortcut:
        w = max(j for j in range(i) if fwd[j][0] == "WAIT" and fwd[j][2] == side)
        e = max(j for j in range(w) if fwd[j] == ("EVENT", fwd[w][1], main))
        between = [r for r in fwd[e:w] if r[0] == "K" and r[2] == main and "spconv" in r[1]]
        assert between, "the fork event of a shortcut stage is not in front of conv1"
    assert sum(r[0] == "WAIT" and r[2] == main for r in fwd) == len(shortcut)
    assert not any(r[0] == "K" and r[2] == side and "spconv" in r[1] for r in fwd)
    # ---- backward
    assert sum(r[0] == "WAIT" and r[2] == main for r in bwd) == 1
    assert bwd[-1][0] == "WAIT" and bwd[-1][2] == main
    wg = [r for r in bwd if r[0] == "K" and "wgrad_tl_kernel" in r[1]]
    assert len(wg) >= 40 and all(r[2] == side for r in wg)
    stem = [i for i, r in enumerate(bwd) if r[0] == "K" and "stem_wgrad_kernel" in r[1]]
    assert len(stem) == 1 and bwd[stem[0]][2] == main
    reduces = [i for i, r in enumerate(bwd) if r[0] == "K" and "wgrad_tl_reduce_batch_kernel" in r[1]]
    assert reduces and all(r < stem[0] for r in reduces) and all(bwd[r][2] == side for r in reduces)
    # the shortcut stages' backward (batch norm + 1x1 input gradient) is on the main stream
    assert not any(r[0] == "K" and r[2] == side and ("dense_kernel" in r[1] or "bn_bwd" in r[1] or "col_reduce" in r[1]) for r in bwd)
