"""Loader of the whole-model fixtures produced by the REFERENCE's CPU inference path (tests/golden/gen_model_fixtures.py,
tests/golden/ref_model_*.npz) and the comparison rules shared by the CPU (oracle) and GPU (HIP engine) tests.

Tolerance: the reference CPU path runs F32 activations over the checkpoint's weights (ggml; RMS eps 1e-6),
the GPU path rounds every op boundary to F16 (RMS eps 1e-5, appendix A5/A14), so logits are compared by cosine >= 0.999
and max |delta| <= LOGIT_MAD (logits here reach |8|, where one F16 ulp is 0.0078), and a greedy id must equal the
reference's whenever the reference's own top-2 gap exceeds LOGIT_TOL (a smaller gap is a tie at this precision); excused
steps are counted and bounded.
"""
import glob
import json
import os

import numpy as np

from tests import engine_fixtures as fx

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOGIT_TOL = 0.03
LOGIT_MAD = 0.05
MIN_COS = 0.999
MAX_EXCUSED_FRACTION = 0.15


def names():
    return sorted(os.path.basename(p)[len("ref_model_"):-4] for p in glob.glob(os.path.join(GOLDEN, "ref_model_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN, "ref_model_%s.npz" % name))
    fxt = {k: z[k] for k in z.files}
    fxt["shape"] = json.loads(str(fxt["shape"]))
    for k in ("seed", "ctx"):
        fxt[k] = int(fxt[k])
    fxt["std"] = float(fxt["std"])
    fxt["shared_classifier"] = bool(fxt["shared_classifier"])
    fxt["fmt"] = str(fxt.get("fmt", "llama2.c"))          # "safetensors": HF names under "model.", config.json hyper-parameters
    fxt["qk_order"] = int(fxt.get("qk_order", 0))        # ModelSpec qk_column_order the reference was run with
    return fxt


def rope_order(fxt):
    """The worker's RoPE pairing for the fixture's qk_column_order (unary_tensor_opr.h:661-735: 2 = (c, c + dims/2))."""
    return 2 if fxt["qk_order"] == 2 else 1


def write_model_dir(d, fxt, **kw):
    """The model directory of the fixture in ITS checkpoint format, for the HIP engine (tests/engine_fixtures.write_model_dir)."""
    return fx.write_model_dir(d, fmt=fxt["fmt"], ctx=fxt["ctx"], s=fxt["shape"], seed=fxt["seed"], std=fxt["std"],
                              shared_classifier=fxt["shared_classifier"], qk_order=fxt["qk_order"], **kw)


def weights(fxt):
    """The checkpoint the reference was run on, regenerated from (shape, seed, std)."""
    return fx.make_weights(fxt["shape"], fxt["seed"], fxt["std"], shared_classifier=fxt["shared_classifier"])


def cos_mad(a, b):
    a = np.asarray(a, np.float32).ravel(); b = np.asarray(b, np.float32).ravel()
    cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    return cos, float(np.abs(a - b).max())


def masked_argmax(logits, excluded):
    lg = np.asarray(logits, np.float32).copy()
    lg[np.asarray(excluded, np.int64)] = -np.inf
    return int(np.argmax(lg))


def check_run(fxt, prefill_logits, step_rows, what):
    """prefill_logits [P][vocab], step_rows = list of [vocab] logits of the teacher-forced decode steps (the reference's
    own tokens are fed back, so one near-tie cannot derail the rest of the comparison)."""
    cos, mad = cos_mad(prefill_logits, fxt["prefill_logits"])
    assert cos >= MIN_COS and mad <= LOGIT_MAD, "%s prefill logits: cos %.6f, max |d| %.4f" % (what, cos, mad)
    rows = [np.asarray(prefill_logits)[-1]] + list(step_rows)
    ref_rows = [fxt["prefill_logits"][-1]] + list(fxt["step_logits"][:len(step_rows)])
    excused, worst_cos, worst_mad = 0, 1.0, 0.0
    for s, (row, ref) in enumerate(zip(rows, ref_rows)):
        cos, mad = cos_mad(row, ref)
        worst_cos, worst_mad = min(worst_cos, cos), max(worst_mad, mad)
        assert cos >= MIN_COS and mad <= LOGIT_MAD, "%s step %d logits: cos %.6f, max |d| %.4f" % (what, s, cos, mad)
        tok = masked_argmax(row, fxt["excluded_ids"])
        if fxt["top2_gap"][s] > LOGIT_TOL:
            assert tok == int(fxt["tokens"][s]), "%s step %d: token %d, reference %d (gap %.4f)" % (
                what, s, tok, int(fxt["tokens"][s]), float(fxt["top2_gap"][s]))
        elif tok != int(fxt["tokens"][s]):
            excused += 1
    assert excused <= MAX_EXCUSED_FRACTION * len(rows), "%s: %d of %d steps excused as ties" % (what, excused, len(rows))
    return dict(steps=len(rows), excused=excused, worst_cos=worst_cos, worst_mad=worst_mad)
