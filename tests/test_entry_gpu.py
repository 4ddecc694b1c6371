"""GPU: the DDP entry point end to end on synthetic features (single process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("hip_graph", [0, 1])
def test_main_retrieval_synthetic_trains_and_evaluates(tmp_path, hip_graph):
    cmd = [sys.executable, os.path.join(ROOT, "main_retrieval.py"), "--do_train", "1", "--synthetic", "--batch_size", "32",
           "--num_neighbors", "8", "--mb_batch", "2", "--epochs", "1", "--synthetic_train", "256", "--synthetic_test", "200",
           "--n_display", "4", "--save_model", "--output_dir", str(tmp_path), "--hip_graph", str(hip_graph)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "memory bank: 64 samples" in r.stdout
    assert "text->video R@1" in r.stdout and "loss" in r.stdout
    losses = [float(l.split(" loss ")[1].split()[0]) for l in r.stdout.splitlines() if " loss " in l]
    assert all(x == x and x < 1e4 for x in losses), losses           # finite
    assert os.path.exists(os.path.join(str(tmp_path), "pytorch_model.bin.0"))
