#!/usr/bin/env python
"""Counterpart of the reference's tools/convert-pretrained-nat-model-to-d2.py (same command line):

    ./convert-pretrained-nat-model-to-d2.py dinat_large_in22k_224.pth dinat_large_in22k_224.pkl

wraps the WHOLE loaded state dict (NAT / DiNAT releases have no "model" level) with "matching_heuristics": True, so that
`patch_embed.*` / `levels.*` keys are matched to `backbone.*` by suffix on load (uenc/checkpoint.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if __name__ == "__main__":
    from uenc.checkpoint import convert_pretrained_nat_model_to_d2
    convert_pretrained_nat_model_to_d2(sys.argv[1], sys.argv[2])
