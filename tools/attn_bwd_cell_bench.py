"""Probe (round 4): one backward attention step with the attention cell's backward inside (gt_attn_bwd_step_cell) against the two
launches it replaces (gt_attn_bwd_step + gt_cell_bwd_query_c), stand-alone, per memory width E and length mix."""
import ctypes, os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gantron_amd import ops
from gantron_amd._lib import lib, STRUCTS
dev = torch.device('cuda:0')
REPS = 200
A, Fn, Kc, H, Apad = 128, 32, 31, 1024, 128


def g(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def bench(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


for (B, Ti, E, mix) in ((64, 187, 1024, 'lj'), (64, 187, 1096, 'lj'), (64, 187, 1096, 'vesus'), (128, 187, 1096, 'vesus')):
    Bp = B
    d = ops.attn_dims(B, Ti, E, A, Fn, Kc)
    nch = int(lib().gt_attn_chunks(Ti))
    if mix == 'lj':
        lens = torch.randint(60, Ti + 1, (B,), generator=torch.Generator().manual_seed(1)); lens[0] = Ti
    else:
        lens = torch.cat((torch.randint(60, Ti + 1, (B // 2,), generator=torch.Generator().manual_seed(1)),
                          torch.randint(3, 39, (B - B // 2,), generator=torch.Generator().manual_seed(2)))); lens[0] = Ti
    lens32 = lens.to(torch.int32).to(dev)
    v = g(1, A, seed=3, scale=0.5).to(dev)
    memory = g(B, Ti, E, seed=4, scale=0.5).to(dev)
    w = torch.softmax(g(B, Ti, seed=20), 1).to(dev)
    u = torch.tanh(g(B, Ti, A, seed=30)).to(dev)
    dctx_dir = g(B, E, seed=40, scale=0.2).to(dev)
    S1 = 10
    dslab = g(S1, B, E, seed=50, scale=0.2).to(dev)
    ph = g(B, Ti, 64, seed=51).to(dev)
    gates = torch.sigmoid(g(Bp, 4 * H, seed=70)).to(dev)
    cst = g(2, Bp, H, seed=85).to(dev)
    keep = (torch.rand(Bp, H, generator=torch.Generator().manual_seed(86)) >= 0.1).to(torch.uint8).to(dev)
    dha = g(Bp, H, seed=87, scale=0.2).to(dev)
    dhc = g(S1, Bp, H, seed=88, scale=0.2).to(dev)
    Wq = g(A, H, seed=89, scale=0.1).to(dev)
    dwcat = torch.zeros(2, 2, B, Ti, device=dev); dcum = torch.zeros(B, Ti, device=dev)
    dq = torch.zeros(Bp, Apad, device=dev); de = torch.zeros(B, Ti, device=dev); dctx_out = torch.zeros(B, E, device=dev)
    sums = torch.zeros(B, nch, device=dev); sums_prev = torch.zeros(B, nch, device=dev)
    dq2 = g(Bp, Apad, seed=90, scale=0.1).to(dev); dwc2 = torch.zeros(2, B, Ti, device=dev)
    dG = torch.zeros(Bp, 4 * H, device=dev); dc = torch.zeros(Bp, H, device=dev)
    cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    sl = ops.slabs(dslab, S1, B * E, E)
    a_sl, c_sl = ops.slabs(dha, 1, 0, H), ops.slabs(dhc, S1, Bp * H, H)
    common = (d, sl, ops.NO_SLABS, dctx_dir, E, memory, E, lens32, dwcat[1], dcum, None, dctx_out, E, w, u, v, ph,
              sums_prev, dwc2, sums, dq, Apad, dwcat[0], de)
    launch = [0]

    def fused():
        launch[0] += 1
        c = STRUCTS['GtCellBwdTail']()
        c.dh_a, c.dh_c = a_sl, c_sl
        c.dq2, c.Wq, c.ldwq = ops.ptr(dq2), ops.ptr(Wq), H
        c.dc_next, c.gates, c.c_new, c.c_prev = ops.ptr(dc), ops.ptr(gates), ops.ptr(cst[1]), ops.ptr(cst[0])
        c.keep, c.p_drop, c.dgates, c.dc_prev, c.H = ops.ptr(keep), 0.1, ops.ptr(dG), ops.ptr(dc), H
        c.row_cnt, c.launch = ops.ptr(cnt), launch[0]
        ops.call('gt_attn_bwd_step_cell', *common, ctypes.byref(c))

    def two():
        ops.call('gt_attn_bwd_step', *common)
        ops.call('gt_cell_bwd_query_c', a_sl, c_sl, dq, Apad, sums, nch, dq2, Wq, H, dc, gates, cst[1], cst[0], keep, 0.1, dG, dc, B, Bp, H, A)

    def first():
        ops.call('gt_attn_bwd_step', *common)

    t2, t1, tf = bench(two), bench(first), bench(fused)
    print(f'B={B} Ti={Ti} E={E} lengths {mix:5s}: attention step alone {t1:6.2f} us, + cell launch {t2:6.2f} us, one launch with the cell inside {tf:6.2f} us', flush=True)
