"""Golden vectors for multi-sentence retrieval (several captions per video): the reference's own
RetrievalMetrics.tensor_text_to_video_metrics / tensor_video_to_text_sim / compute_metrics
(utils/metrics.py:39-148) on the padded tensor that evaluator.py:236-250 builds, run HERE where
/root/reference exists; exits quietly anywhere else.  Writes tests/golden/multi_sentence.npz:
the sentence x video matrix, the cut-off points and both metric dictionaries.

The matrix has no exact ties inside a row (the reference's argsort is not stable: what it does with
ties is unspecified), but ties ACROSS the sentences of a group and NaN scores are planted: those the
reference defines (max over the group, NaN -> -inf).
"""
import os
import sys

import numpy as np

from capture_golden import OUT, REF, ROOT, _import_reference

KEYS_T = ("R1", "R5", "R10", "R50", "MedianR", "MeanR", "Std_Rank", "MR")
KEYS_V = ("R1", "R5", "R10", "R50", "MR", "MeanR")


def problem():
    from neighborretr_amd import synth
    V = 37
    sizes = 1 + (np.arange(V) * 7) % 5                           # 1..5 sentences per video
    ends = np.cumsum(sizes)
    Ns = int(ends[-1])
    S = (synth.normal(77, "multi/S", (Ns, V)) * 0.1).astype(np.float32)
    group = np.searchsorted(ends, np.arange(Ns), side="right")
    S[np.arange(Ns), group] += 0.12                              # own video usually near the top, not always first
    S[3, 5 if group[3] != 5 else 6] = np.nan                     # a NaN score off the own column: ranks first
    S[10, group[10]] = np.nan                                    # a NaN own score: that sentence is not ranked
    S[ends[4] - 1, 9] = S[ends[4] - 2, 9]                        # equal scores in one group's column (max is the same)
    return S, (ends - 1).tolist()


def main():
    if not os.path.isdir(REF):
        print("reference checkout not present; nothing to capture")
        return 0
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nr_oracle as O
    _, RetrievalMetrics = _import_reference()
    S, cut = problem()
    padded = O.pad_sentence_groups(S, cut)
    t2v = RetrievalMetrics.tensor_text_to_video_metrics(padded.copy())
    v2t = RetrievalMetrics.compute_metrics(RetrievalMetrics.tensor_video_to_text_sim(padded.copy()))
    mine_t, mine_v = O.multi_sentence_metrics(S, cut)
    for k in KEYS_T:
        assert abs(t2v[k] - mine_t[k]) < 1e-5 * max(1.0, abs(t2v[k])), (k, t2v[k], mine_t[k])   # the reference rounds R@K to fp32
    assert v2t["cols"] == mine_v["cols"]
    np.savez_compressed(os.path.join(OUT, "multi_sentence.npz"), S=S, cut_off_points=np.array(cut),
                        t2v=np.array([t2v[k] for k in KEYS_T]), v2t=np.array([v2t[k] for k in KEYS_V]),
                        v2t_cols=np.array(v2t["cols"]))
    print(f"[multi_sentence] oracle == reference; {S.shape[0]} sentences x {S.shape[1]} videos, "
          f"t2v R1={t2v['R1']:.2f} MedianR={t2v['MedianR']}, v2t R1={v2t['R1']:.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
