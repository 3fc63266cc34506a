"""Resolve the reference's shipped config chain into a flat key -> value fixture (build container only).

TEST INFRASTRUCTURE.  Run:  python -m oracle.make_cfg_fixture
Reads /root/reference/configs/cityscapes/swin/unified_encoder_cityscapes.yaml (its `_BASE_` chain ->
oneformer_R50_bs16_90k.yaml -> Base-Cityscapes-UnifiedSegmentation.yaml, and the `!!python/object/apply:eval` tag) through the
product's own config loader on top of the product's defaults, and writes every resolved key as data to
tests/golden/cfg_cityscapes_swin_t.json.  The YAML text itself does not travel; the GPU box rebuilds the cfg from these values
(BASELINE configs[0]: the DefaultPredictor counterpart on the reference's Swin-T config, demo/defaults.py:51-61, 157-158).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "uni-encoder-code_amd"))
REF_CFG = "/root/reference/configs/cityscapes/swin/unified_encoder_cityscapes.yaml"


def flatten(node, prefix=""):
    out = {}
    for k, v in node.items():
        if isinstance(v, dict):
            out.update(flatten(v, prefix + k + "."))
        else:
            out[prefix + k] = list(v) if isinstance(v, tuple) else v
    return out


def main():
    import model  # noqa: F401
    from uenc.config import add_common_config, add_dinat_config, add_swin_config, add_uni_encoder_config
    from uenc.d2 import get_cfg
    cfg = get_cfg()
    add_common_config(cfg); add_swin_config(cfg); add_dinat_config(cfg); add_uni_encoder_config(cfg)
    cfg.merge_from_file(REF_CFG)
    flat = flatten(cfg)
    path = os.path.join(ROOT, "tests", "golden", "cfg_cityscapes_swin_t.json")
    with open(path, "w") as f:
        json.dump({"source": "configs/cityscapes/swin/unified_encoder_cityscapes.yaml (+ _BASE_ chain), resolved values only",
                   "generator": "python -m oracle.make_cfg_fixture", "cfg": flat}, f, indent=1, sort_keys=True)
    print(f"{path}: {len(flat)} keys")


if __name__ == "__main__":
    main()
