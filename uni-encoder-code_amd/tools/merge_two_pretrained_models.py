#!/usr/bin/env python
"""Counterpart of the reference's tools/merge_two_pretrained_models.py (same command line):

    ./merge_two_pretrained_models.py swin_tiny_patch4_window7_224.pkl r18.pkl swin_tiny_patch4_window7_224_r18.pkl

the first wrapper's state dict updated with the second's, written with "matching_heuristics": True (uenc/checkpoint.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if __name__ == "__main__":
    from uenc.checkpoint import merge_two_pretrained_models
    merge_two_pretrained_models(sys.argv[1], sys.argv[2], sys.argv[3])
