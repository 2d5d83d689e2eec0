"""The inert pickle reader: reads the reference's shipped LoRA checkpoint, executes nothing."""
import os
import pickle

import numpy as np
import pytest


def test_reads_shipped_lora_checkpoint(golden_dir):
    from clipfs import safe_pkl
    d = safe_pkl.load(os.path.join(golden_dir, "lora_weights.pkl"))
    assert d["metadata"] == {"r": 4, "alpha": 1, "encoder": "both", "params": ["q", "k", "v"], "position": "all"}
    w = d["weights"]
    assert len(w) == 24
    n = 0
    for i in range(24):
        width = 512 if i < 12 else 768  # text blocks first, then vision (apply_lora order)
        for p in ("q_proj", "k_proj", "v_proj"):
            a, b = w[f"layer_{i}"][p]["w_lora_A"], w[f"layer_{i}"][p]["w_lora_B"]
            assert a.shape == (4, width) and b.shape == (width, 4) and a.dtype == np.float32
            n += a.size + b.size
    assert n == 368640


def test_roundtrip_of_own_files(tmp_path):
    from clipfs import safe_pkl
    obj = {"weights": {"layer_0": {"q_proj": {"w_lora_A": np.arange(12, dtype=np.float32).reshape(3, 4),
                                              "w_lora_B": np.ones((4, 3), np.float64)}}},
           "metadata": {"r": 3, "alpha": 1, "params": ["q"], "ok": True, "none": None, "f": 0.5}}
    p = tmp_path / "x.pkl"
    with open(p, "wb") as f:
        pickle.dump(obj, f, protocol=4)
    got = safe_pkl.load(str(p))
    assert got["metadata"] == obj["metadata"]
    assert np.array_equal(got["weights"]["layer_0"]["q_proj"]["w_lora_A"], obj["weights"]["layer_0"]["q_proj"]["w_lora_A"])
    assert got["weights"]["layer_0"]["q_proj"]["w_lora_B"].dtype == np.float64


class _Evil:
    def __reduce__(self):
        import os
        return (os.system, ("echo pwned > /tmp/clipfs_pwned",))


def test_refuses_code_execution(tmp_path):
    from clipfs import safe_pkl
    if os.path.exists("/tmp/clipfs_pwned"):
        os.remove("/tmp/clipfs_pwned")
    blob = pickle.dumps({"x": _Evil()}, protocol=4)
    with pytest.raises(safe_pkl.UnsafePickleError):
        safe_pkl.loads(blob)
    assert not os.path.exists("/tmp/clipfs_pwned")
    with pytest.raises(safe_pkl.UnsafePickleError):
        safe_pkl.loads(pickle.dumps(np.array([object()], dtype=object), protocol=4))
