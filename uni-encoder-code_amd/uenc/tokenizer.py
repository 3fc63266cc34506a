"""Task-prompt token ids.

The reference tokenises `"The task is {panoptic|instance|semantic}"` with a CLIP BPE tokenizer
(model/data/tokenizer.py:87-193, vocabulary file of 16e6 merges) on every forward
(model/oneformer_model.py:249-251).  On the segmentation path only these three prompts ever occur,
so their ids are constants (verified against the reference tokenizer by oracle/make_golden.py and
pinned in tests/golden/task_tokens.npz); the BPE vocabulary never needs to ship.  Any other text is
rejected loudly rather than tokenised differently.
"""
import torch

_SOT, _EOT = 49406, 49407
_TASK_IDS = {
    "the task is panoptic": [_SOT, 518, 10549, 533, 1072, 24755, _EOT],
    "the task is semantic": [_SOT, 518, 10549, 533, 29119, 1550, _EOT],
    "the task is instance": [_SOT, 518, 10549, 533, 34572, _EOT],
}


class Tokenize:
    def __init__(self, tokenizer=None, max_seq_len=77, truncate=True):
        self.max_seq_len = max_seq_len

    def __call__(self, texts):
        single = isinstance(texts, str)
        if single:
            texts = [texts]
        out = torch.zeros(len(texts), self.max_seq_len, dtype=torch.long)
        for i, t in enumerate(texts):
            key = " ".join(t.lower().split())
            if key not in _TASK_IDS:
                raise KeyError(f"only the three task prompts are tokenised on the hot path, got {t!r}")
            ids = _TASK_IDS[key]
            out[i, : len(ids)] = torch.tensor(ids)
        return out[0] if single else out
