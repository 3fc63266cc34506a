import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "uni-encoder-code_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# oracle/ref_loader.py seeds sys.modules with name holders (`model`, `detectron2`, `timm`, `fvcore`, `cv2`, ...) so that the reference's
# files import in the build container.  They must not outlive the test that needed them: the product's own `model` facade and its
# "is detectron2 installed?" probe would otherwise see them, depending on the order the test files run in.
_STUB_ROOTS = ("model", "detectron2", "timm", "fvcore", "cv2", "skimage", "matplotlib", "natten")


@pytest.fixture(autouse=True)
def _isolate_reference_stubs():
    before = {k: v for k, v in sys.modules.items() if k.split(".")[0] in _STUB_ROOTS}
    yield
    rl = sys.modules.get("oracle.ref_loader")
    if rl is None or not getattr(rl, "_installed", False):
        return
    for k in [k for k in sys.modules if k.split(".")[0] in _STUB_ROOTS and k not in before]:
        del sys.modules[k]
    sys.modules.update(before)
    rl._installed = False


def load_golden(name):
    """npz fixture -> dict of torch tensors (strings/ints stay numpy)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" and a.dtype != np.uint16 else a
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()


# ---- parity figures: every GPU parity test records the numbers it computed (relative errors, sign agreement, gradient
# cosines, envelope values), not only whether they passed.  Written at session end to $UENC_PARITY_OUT or, by default,
# gpurun_out/parity.json (which gpurun merges back); the copy a round wants judged is committed as profiles/rNN_parity.json.
_PARITY = {}


def record_parity(test: str, **figures):
    ent = _PARITY.setdefault(test, {})
    for k, v in figures.items():
        ent[k] = float(v) if isinstance(v, (int, float, np.floating)) or (torch.is_tensor(v) and v.numel() == 1) else v


MASK_ABS_TOL = 5e-2          # bf16 mode, abs error of the `pred_masks` logits


def mask_band_figures(pred, ref, tol=MASK_ABS_TOL):
    """Where the mask-sign disagreements sit, in the contract's own unit (`pred_masks` logits within 5e-2 ABS).
    A mask pixel's sign is a thresholded logit: a product logit within `tol` of the reference's may legitimately land on the other
    side of 0 iff |reference logit| <= tol.  Returns abs-error statistics, the number of sign flips, the share of them inside that
    band, and the sign agreement over the pixels OUTSIDE the band (where no in-tolerance error can flip a sign)."""
    p, r = pred.detach().double().cpu().reshape(-1), ref.detach().double().cpu().reshape(-1)
    err = (p - r).abs()
    flips = (p > 0) != (r > 0)
    band = r.abs() <= tol
    nflip = int(flips.sum())
    outside = ~band
    return {"abs_err_max": float(err.max()), "abs_err_mean": float(err.mean()), "abs_err_p999": float(err.kthvalue(max(1, int(0.999 * err.numel()))).values),
            "ref_abs_mean": float(r.abs().mean()), "pixels": int(p.numel()), "sign_flips": nflip,
            "mask_sign_agreement": 1.0 - nflip / p.numel(), "share_of_pixels_in_band": float(band.double().mean()),
            "flips_inside_band_share": float((flips & band).sum()) / max(nflip, 1),
            "flips_outside_band": int((flips & outside).sum()),
            "sign_agreement_outside_band": 1.0 - float((flips & outside).sum()) / max(int(outside.sum()), 1),
            "max_ref_abs_at_a_flip": float(r.abs()[flips].max()) if nflip else 0.0, "band_abs_tol": tol}


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    path = os.environ.get("UENC_PARITY_OUT", os.path.join(ROOT, "gpurun_out", "parity.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        old = {}
        if os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
        old.update(_PARITY)
        with open(path, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass
