import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/uni-encoder-code_amd'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_golden
from test_model_gpu import _head_modules, rel
import uenc.modeling
from oracle import torch_ref as T, fill
g = load_golden("transformer_decoder")
ch = {"res2": 96, "res3": 192, "res4": 384, "res5": 768}
_, dec = _head_modules(None, ch)
with torch.no_grad():
    o = dec([g["ms0"].cuda(), g["ms1"].cuda(), g["ms2"].cuda()], g["mask_features"].cuda(), g["tasks"].cuda())
for i, a in enumerate(o["aux_outputs"]):
    print(i, 'logits rel', rel(a["pred_logits"], g[f"aux{i}_logits"]), 'masks rel', rel(a["pred_masks"], g[f"aux{i}_masks"].float()),
          'sign agree', float(((a["pred_masks"].cpu()>0)==(g[f"aux{i}_masks"].float()>0)).float().mean()))
print('final', rel(o["pred_logits"], g["pred_logits"]), rel(o["pred_masks"], g["pred_masks"]))
# sensitivity: oracle with bf16-rounded weights
sd = fill.state_dict_for({k: s for k, s in T.head_param_shapes(T.HeadCfg(), ch).items() if "predictor" in k})
sd16 = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
with torch.no_grad():
    o2 = T.transformer_decoder([g["ms0"], g["ms1"], g["ms2"]], g["mask_features"], g["tasks"], sd16, T.HeadCfg())
for i, a in enumerate(o2["aux_outputs"]):
    print('oracle-bf16w', i, rel(a["pred_logits"], g[f"aux{i}_logits"]), rel(a["pred_masks"], g[f"aux{i}_masks"].float()))
print('oracle-bf16w final', rel(o2["pred_logits"], g["pred_logits"]), rel(o2["pred_masks"], g["pred_masks"]))
