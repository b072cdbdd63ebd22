#!/usr/bin/env python
"""In-kernel timeline of the event-level attention FORWARD (mh_attn_fwd_timeline, A/B library): lane 0 of every wave of sixteen
workgroups stamps s_memtime at the seams of each key tile's segments; this prints the average shader cycles per segment for every
wave, the spread between the four waves of a workgroup at the barrier, and the tile period.  B=16, H=16, bf16.
usage: attn_timeline.py [S] [lazy]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from midi_model_amd import ops  # noqa: E402

_AB = ops.ab_library()
_AB.__enter__()
from midi_model_amd.lib import lib  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lazy = int(sys.argv[2]) if len(sys.argv) > 2 else 0
first = int(sys.argv[3]) if len(sys.argv) > 3 else 8   # first recorded key tile of each workgroup's loop
B, H = 16, 16
D = H * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn((B * S, 3 * D), device="cuda", generator=g).to(torch.bfloat16)
o = torch.empty((B * S, D), device="cuda", dtype=torch.bfloat16)
Sp = (S + 63) // 64 * 64
lse = torch.zeros(B * H * Sp, device="cuda")
NWG, NW, NT, NS = 16, 4, 32, 9
nqt = (S + 127) // 128
GRID = nqt * 8 * ((B * H + 7) // 8)
st = torch.zeros(NWG * NW * NT * NS + GRID * 8 + 1, dtype=torch.int32, device="cuda")
st[-1] = first
stream = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    lib().call("mh_attn_fwd_timeline", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, S, H, 0.125, lazy, st.data_ptr(), stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    lib().call("mh_attn_fwd_timeline", qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), B, S, H, 0.125, lazy, st.data_ptr(), stream)
e1.record()
torch.cuda.synchronize()
t_tl = e0.elapsed_time(e1) / 5 * 1e3
ops.set_option("attn_v3", 127 | (128 if lazy else 0))
for _ in range(2):
    ops.attn_fwd(qkv, o, lse, B, S, H, 0.125)
e0.record()
for _ in range(5):
    ops.attn_fwd(qkv, o, lse, B, S, H, 0.125)
e1.record()
torch.cuda.synchronize()
t_prod = e0.elapsed_time(e1) / 5 * 1e3
print(f"# attention forward B={B} H={H} S={S} lazy={lazy}: production form {t_prod:.1f} us, instrumented build {t_tl:.1f} us per launch")
a = (st[:NWG * NW * NT * NS].cpu().view(NWG, NW, NT, NS).to(torch.int64) & 0xffffffff)
names = ["LDS-DMA issue (4 requests)", "K reads issue (8 b128)", "S MFMAs + V^T reads issue", "softmax arithmetic",
         "wait V^T fragments", "P V MFMAs issue", "wait next tile's DMA (vmcnt)", "barrier"]
print(f"# shader cycles per segment, averaged over the recorded tiles ({first} .. {first + 31} of the workgroup's loop); one line per wave")
print("# wg wave  " + "  ".join(f"{n[:14]:>14s}" for n in names) + "    tile period   valid tiles")
tot = torch.zeros(len(names))
cnt = 0
for w in range(NWG):
    for v in range(NW):
        s = a[w, v]
        ok = (s[:, 0] != 0) & (s[:, 8] != 0)
        n = int(ok.sum())
        if n < 4:
            continue
        seg = ((s[:, 1:] - s[:, :-1]) & 0xffffffff)[ok].float()        # [n, 8]
        idx = torch.nonzero(ok).flatten()
        per = ((s[idx[1:], 0] - s[idx[:-1], 0]) & 0xffffffff).float()
        per = per[(idx[1:] - idx[:-1]) == 1]
        m = seg.mean(0)
        tot += m
        cnt += 1
        print(f"  {w:2d}  {v:2d}   " + "  ".join(f"{x:14.0f}" for x in m.tolist()) + f"    {per.mean().item():10.0f}   {n:4d}")
print("# mean over waves: " + ", ".join(f"{n} {x:.0f}" for n, x in zip(names, (tot / max(cnt, 1)).tolist())) +
      f"  | sum {float(tot.sum() / max(cnt, 1)):.0f} cycles per tile")
# skew at the barrier: arrival (stamp 7) of the four waves of a workgroup at the same tile
for w in range(min(NWG, 4)):
    s = a[w]
    ok = (s[:, :, 0] != 0).all(0) & (s[:, :, 8] != 0).all(0)
    if int(ok.sum()) < 4:
        continue
    arr = s[:, ok, 7].float()
    rel = s[:, ok, 8].float()
    print(f"# wg {w}: barrier arrival spread (max - min over the 4 waves) mean {float((arr.max(0).values - arr.min(0).values).mean()):.0f} cycles; "
          f"release - last arrival mean {float((rel.min(0).values - arr.max(0).values).mean()):.0f}")

# ---- residency census: every workgroup's [start, end] on its CU ------------------------------------------------------------
c = (st[NWG * NW * NT * NS:-1].cpu().view(GRID, 8).to(torch.int64) & 0xffffffff)
t0 = c[:, 0] | (c[:, 1] << 32)
t1 = c[:, 2] | (c[:, 3] << 32)
hw, xcc, tiles = c[:, 4], c[:, 5] & 0xf, c[:, 6]
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
key = (xcc << 12) | (se << 8) | (sh << 4) | cu
ok = t1 > t0
for x in range(8):
    mk = ok & (xcc == x)
    if int(mk.sum()):
        span = (t1[mk].max() - t0[mk].min()).item()
        print(f"# XCD {x}: {int(mk.sum())} workgroups, span {span} ticks = {span / t_tl:.0f} ticks per us of the instrumented launch; heaviest workgroup "
              f"{int((t1 - t0)[mk].max())} ticks for {int(tiles[mk][(t1 - t0)[mk].argmax()])} tiles")
print(f"# census: {int(ok.sum())} of {GRID} workgroups recorded; distinct (xcc, se, sh, cu) = {len(set(key[ok].tolist()))}; "
      f"kernel span {(t1[ok].max() - t0[ok].min()).item()} ticks of s_memtime; workgroup life mean {float((t1 - t0)[ok].float().mean()):.0f} "
      f"max {int((t1 - t0)[ok].max())}; ticks per key tile (life / tiles) mean {float(((t1 - t0)[ok].float() / tiles[ok].float()).mean()):.0f}")
import collections
per = collections.defaultdict(list)
for k, a0, a1 in zip(key[ok].tolist(), t0[ok].tolist(), t1[ok].tolist()):
    per[k].append((a0, a1))
avg, peak = [], []
for k, iv in per.items():
    ev = sorted([(x, 1) for x, _ in iv] + [(y, -1) for _, y in iv])
    cur = mx = 0
    area = 0
    last = ev[0][0]
    for t, d in ev:
        area += cur * (t - last)
        last = t
        cur += d
        mx = max(mx, cur)
    span = ev[-1][0] - ev[0][0]
    avg.append(area / max(span, 1))
    peak.append(mx)
avg_t, peak_t = torch.tensor(avg), torch.tensor(peak)
print(f"# workgroups resident per CU: time-average mean {float(avg_t.mean()):.2f} (min {float(avg_t.min()):.2f}, max {float(avg_t.max()):.2f}); "
      f"peak mean {float(peak_t.float().mean()):.2f} (min {int(peak_t.min())}, max {int(peak_t.max())}); workgroups per CU mean {GRID / max(len(per), 1):.1f}")
