import sys, torch
sys.path.insert(0, '/root/repo')
import midi_model_amd as mm
from midi_model_amd import ops
tok = mm.MIDITokenizerV2()
first, lo_t, hi_t, _ = tok.grammar_tables()
B, V, Vp = 64, tok.vocab_size, 3456
g = torch.Generator().manual_seed(1)
logits = torch.zeros((B, Vp), dtype=torch.bfloat16)
logits[:, :V] = (3 * torch.randn((B, V), generator=g)).to(torch.bfloat16)
fm = torch.tensor(first, dtype=torch.uint8).cuda()
lo, hi = torch.tensor(lo_t, dtype=torch.int32).cuda(), torch.tensor(hi_t, dtype=torch.int32).cuda()
span, mr = ops.mask_spans(fm.cpu(), lo.cpu(), hi.cpu())
q = torch.empty((B, V)).exponential_(1.0, generator=g).cuda()
lg = logits.cuda()
out = torch.zeros((B, 8), dtype=torch.int64, device='cuda')
for pos, evid in ((1, 3), (4, 6), (7, 3)):
    ev = torch.full((B,), evid, dtype=torch.int64, device='cuda')
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ops.sample_top_p_k(lg, fm, lo, hi, ev, pos, q, out[:, pos], V, 1.0, 0.98, 20, first_span=span, max_range=mr[pos])
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for _ in range(100):
                ops.sample_top_p_k(lg, fm, lo, hi, ev, pos, q, out[:, pos], V, 1.0, 0.98, 20, first_span=span, max_range=mr[pos])
        best = 1e9
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    print(f"pos {pos} (range {mr[pos]}): {10 * best:.2f} us per launch")
