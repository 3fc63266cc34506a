#!/usr/bin/env python
"""Counterpart of the reference's tools/convert-pretrained-model-to-d2.py (same command line):

    ./convert-pretrained-model-to-d2.py swin_tiny_patch4_window7_224.pth swin_tiny_patch4_window7_224.pkl

wraps the `"model"` state dict of a torch checkpoint as {"model", "__author__": "third_party", "matching_heuristics": False}
(uenc/checkpoint.py); the input is read with the weights-only loader.  Use with MODEL.WEIGHTS and INPUT.FORMAT "RGB"."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if __name__ == "__main__":
    from uenc.checkpoint import convert_pretrained_model_to_d2
    convert_pretrained_model_to_d2(sys.argv[1], sys.argv[2])
