"""Dev tool: where does the HIP encoder drift from the fp64 oracle? (resnet50, 128px, B=8)"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import encoder as E
from oracle import prng, sat_oracle as O
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
px = int(sys.argv[2]) if len(sys.argv) > 2 else 128
hp = O.default_hparams(encoder_arch=arch, encoder_dim=32, input_size=px)
torch.manual_seed(3); ref = O.build_encoder(hp)
enc = E.get_encoder(O.default_hparams(encoder_arch=arch, encoder_dim=32, input_size=px)); enc.load_state_dict(ref.state_dict()); enc = enc.cuda().train()
ref64 = copy.deepcopy(ref).double()
img = torch.from_numpy(prng.uniform((8, 3, px, px), 77, 0.0, 1.0))
y32 = ref(img.clone()); dy = torch.from_numpy(prng.uniform(tuple(y32.shape), 78)); y32.backward(dy)
y64 = ref64(img.double().clone()); y64.backward(dy.double())
y = enc(img.cuda()); y.backward(dy.cuda())
rel = lambda a, b: float((a.detach().double().cpu() - b.detach()).norm() / b.detach().norm())
print("fwd: hip %.2e cpu32 %.2e" % (rel(y, y64), rel(y32, y64)))
g32, g64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
for k, p in enc.named_parameters():
    if p.dim() == 4:
        print("%-28s hip %.2e  cpu32 %.2e" % (k, rel(p.grad, g64[k].grad), rel(g32[k].grad, g64[k].grad)))
