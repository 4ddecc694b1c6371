#!/usr/bin/env python
"""How well do the branches of the step overlap?  Times (configs[1] shape, un-profiled):
  * text / video clustering and the local branch alone, each as a single-stream HIP graph;
  * pairs / triples as ONE graph with forked streams (what the step does);
  * the same branches as SEPARATE single-stream graphs replayed on different streams.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neighborretr_amd import head, hip, modeling, ops, synth  # noqa: E402

DEV = "cuda"
B, Nt, Nv, M, K = 128, 24, 12, 512, 20


def capture(fn, stream=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    if stream is None:
        with torch.cuda.graph(g):
            fn()
    else:
        with torch.cuda.graph(g, stream=stream):
            fn()
    return g


def time_it(run, reps=200):
    for _ in range(10):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    m = modeling.NeighborRetr(modeling.default_config(num_neighbors=K))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(7).items()}, strict=False)
    m = m.to(DEV).train()
    p = {k: torch.from_numpy(v).to(DEV) for k, v in synth.make_problem(1002, B, Nt, Nv, M).items()}
    tm, vm = p["text_mask"].float(), p["video_mask"].float()
    btm, bvm = p["mb_mask_t"].float(), p["mb_mask_v"].float()
    with torch.no_grad():
        nz = m._draw_noise(B, Nt, Nv, torch.device(DEV))
        sw_t, sw_v = m.scorer_weights("text_weight_fc"), m.scorer_weights("video_weight_fc")

        def text():
            m._merge_one("text", p["text_feat"], tm, nz["t0"], nz["t1"])

        def video():
            m._merge_one("video", p["video_feat"], vm, nz["v0"], nz["v1"])

        def local():
            pt = ops.prepare_tokens(p["text_feat"], tm, want_lo=True, want_colsum=True)
            pv = ops.prepare_tokens(p["video_feat"], vm, want_lo=True, want_colsum=True)
            pbt = ops.prepare_tokens(p["mb_feat_t"], btm, want_lo=False)
            pbv = ops.prepare_tokens(p["mb_feat_v"], bvm, want_lo=False)
            w_t, _ = head.token_weights(pt, tm, sw_t, B, Nt, hip.PREC_BF16X3, False)
            w_v, _ = head.token_weights(pv, vm, sw_v, B, Nv, hip.PREC_BF16X3, False)
            w_bt, _ = head.token_weights(pbt, btm, sw_t, M, Nt, hip.PREC_BF16, False)
            w_bv, _ = head.token_weights(pbv, bvm, sw_v, M, Nv, hip.PREC_BF16, False)
            ops.local_level(pt, pv, w_t, w_v, B, Nt, B, Nv, hip.PREC_BF16X3, hip.OUT_FULL, False)
            ops.local_level(pt, pbv, w_t, w_bv, B, Nt, M, Nv, hip.PREC_BF16, hip.OUT_ROWSUM, False)
            ops.local_level(pbt, pv, w_bt, w_v, M, Nt, B, Nv, hip.PREC_BF16, hip.OUT_COLSUM, False)

        def tiny_chain(n=50):
            x = torch.zeros(8, 4, device=DEV)
            for _ in range(n):
                ops.loss_finalize(x, 1, 1, 1)

        side = [torch.cuda.Stream() for _ in range(3)]

        def forked(fns):
            def run():
                cur = torch.cuda.current_stream()
                for s, f in zip(side, fns[1:]):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        f()
                fns[0]()
                for s, _ in zip(side, fns[1:]):
                    cur.wait_stream(s)
            return run

        branches = {"text": text, "video": video, "local": local}
        alone = {}
        for k, f in branches.items():
            g = capture(f)
            alone[k] = time_it(g.replay)
            print(f"{k:6s} alone                       : {alone[k]:8.1f} us")
        g = capture(lambda: tiny_chain(50))
        print(f"50 dependent ~4.6us kernels, per node: {time_it(g.replay) / 50:8.2f} us")
        branches["text2"], branches["video2"], branches["local2"] = (lambda: text()), (lambda: video()), (lambda: local())
        alone["text2"], alone["video2"], alone["local2"] = alone["text"], alone["video"], alone["local"]
        for combo in (("text", "text2"), ("video", "video2"), ("local", "local2"), ("text", "video"), ("text", "local"), ("local", "text"), ("text", "video", "local"), ("local", "text", "video")):
            g = capture(forked([branches[c] for c in combo]))
            print(f"one graph, forked {'|'.join(combo):20s}: {time_it(g.replay):8.1f} us   (max alone {max(alone[c] for c in combo):.1f}, sum {sum(alone[c] for c in combo):.1f})")
        # separate single-stream graphs on separate streams
        for combo in (("text", "video"), ("text", "video", "local")):
            streams = [torch.cuda.Stream() for _ in combo]
            graphs = [capture(branches[c]) for c in combo]

            def run():
                cur = torch.cuda.current_stream()
                for s, gr in zip(streams, graphs):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        gr.replay()
                for s in streams:
                    cur.wait_stream(s)
            print(f"separate graphs  {'|'.join(combo):21s}: {time_it(run):8.1f} us")


if __name__ == "__main__":
    main()
