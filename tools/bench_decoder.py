"""Dev tool: time the HIP decoder fwd+loss+bwd at a config's decoder shapes (synthetic inputs)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sat_amd  # noqa
from sat_amd import model as M
from oracle import prng, sat_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=128); ap.add_argument("--R", type=int, default=5); ap.add_argument("--T", type=int, default=22)
ap.add_argument("--L", type=int, default=49); ap.add_argument("--D", type=int, default=512); ap.add_argument("--V", type=int, default=6400)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--ragged", action="store_true"); ap.add_argument("--bf16", action="store_true")
a = ap.parse_args()
hp = O.default_hparams(vocab_size=a.V, encoder_dim=a.D, embed_dim=256, attention_dim=128, decoder_dim=512)
dec = M.SATDecoder(hp).cuda()
if a.bf16: dec.sat_precision = "bf16"
ann = torch.from_numpy(prng.uniform((a.B, a.L, a.D), 1, 0.0, 2.0)).cuda().requires_grad_()
caps, lengths = prng.captions(a.B, a.R, a.T, a.V, 2, ragged=a.ragged, min_len=8)
caps, lengths = torch.from_numpy(caps).cuda(), torch.from_numpy(lengths)
def step():
    res = dec.train_decode(ann, caps, lengths, 1.0)
    (res["ce"] + res["ds"]).backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(a.steps): step()
torch.cuda.synchronize(); dt = (time.time() - t0) / a.steps
print("decoder fwd+loss+bwd: %.2f ms/step  (%d captions -> %.0f captions/s decoder-only)" % (dt * 1e3, a.B * a.R, a.B * a.R / dt))
