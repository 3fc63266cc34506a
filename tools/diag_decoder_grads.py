import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/uni-encoder-code_amd'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_golden
from test_model_gpu import _head_modules, rel
import model
from oracle import torch_ref as T, fill
g = load_golden("transformer_decoder")
ch = {"res2": 96, "res3": 192, "res4": 384, "res5": 768}
_, dec = _head_modules(None, ch)
sd = fill.state_dict_for({k: s for k, s in T.head_param_shapes(T.HeadCfg(), ch).items() if "predictor" in k})
sd = {k: v.requires_grad_() for k, v in sd.items()}
feats = [g["ms0"], g["ms1"], g["ms2"]]
mf = g["mask_features"].clone().requires_grad_()
o = T.transformer_decoder(feats, mf, g["tasks"], sd, T.HeadCfg())
loss = T.synthetic_loss(o); loss.backward()
dec.forced_attn_masks = [m.cuda() for m in o["attn_masks"]]
mfg = g["mask_features"].cuda().requires_grad_()
o2 = dec([f.cuda() for f in feats], mfg, g["tasks"].cuda())
loss2 = T.synthetic_loss(o2); loss2.backward()
print('loss', float(loss), float(loss2))
rows = []
for k, p in dec.named_parameters():
    want = sd["sem_seg_head.predictor." + k].grad
    if p.grad is None: print('NO GRAD', k); continue
    got = p.grad.cpu()
    cos = float(torch.nn.functional.cosine_similarity(got.reshape(-1), want.reshape(-1), dim=0))
    rows.append((float(got.norm()/want.norm()), cos, k))
rows.sort()
for r in rows[:25]: print('%.4f %.4f %s' % r)
print('...')
for r in rows[-8:]: print('%.4f %.4f %s' % r)
print('dmf ratio', float(mfg.grad.cpu().norm()/mf.grad.norm()), 'cos', float(torch.nn.functional.cosine_similarity(mfg.grad.cpu().reshape(-1), mf.grad.reshape(-1), dim=0)))
