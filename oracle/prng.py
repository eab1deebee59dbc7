"""TEST INFRASTRUCTURE ONLY -- counter-based PRNG (splitmix64 finaliser over an
index) so fixtures and parity tests regenerate bit-identical inputs on any
machine, independent of torch/numpy generator implementations."""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)


def _mix(z):
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def bits(n, seed):
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        return _mix((idx + np.uint64(1)) * _G + _mix(np.uint64(seed) * _G + np.uint64(0x1234567)))


def uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 array, each value lo + (hi-lo) * k / 2**24 with k a 24-bit integer."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (bits(n, seed) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def integers(shape, seed, lo, hi):
    """int64 array uniform in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    return (lo + (bits(n, seed) >> np.uint64(11)) % np.uint64(hi - lo)).astype(np.int64).reshape(shape)


def decoder_state(hp, seed):
    """Decoder state dict (reference key names) as numpy arrays with
    fan-in scaling similar to torch's default initialisers."""
    D, m, A, n, V, K = hp.encoder_dim, hp.embed_dim, hp.attention_dim, hp.decoder_dim, hp.vocab_size, hp.decoder_layers
    shapes = [("embedding.weight", (V, m), 1.0),
              ("init_lstm.factorize.weight", (m, D), D), ("init_lstm.factorize.bias", (m,), D),
              ("init_lstm.init.weight", (2 * n * K, m), m), ("init_lstm.init.bias", (2 * n * K,), m)]
    for k in range(K):
        cin = m + D if k == 0 else n
        shapes += [("lstm.weight_ih_l%d" % k, (4 * n, cin), n), ("lstm.weight_hh_l%d" % k, (4 * n, n), n),
                   ("lstm.bias_ih_l%d" % k, (4 * n,), n), ("lstm.bias_hh_l%d" % k, (4 * n,), n)]
    shapes += [("attention.encoder_att.weight", (A, D), D), ("attention.decoder_att.weight", (A, n), n),
               ("attention.f_att.weight", (1, A), A), ("beta.0.weight", (D, n), n), ("beta.0.bias", (D,), n),
               ("output.hidden.weight", (m, n), n)]
    if hp.deep_output:
        shapes.append(("output.context.weight", (m, D), D))
    tied = hp.weight_tying and hp.deep_output
    if not tied:
        shapes.append(("output.output.weight", (V, m), m))
    if not hp.weight_tying:
        shapes.append(("output.output.bias", (V,), m))
    sd = {}
    for i, (name, shape, fan) in enumerate(shapes):
        b = 1.0 / np.sqrt(fan)
        sd[name] = uniform(shape, seed * 1000 + i, -b, b)
    sd["embedding.weight"][0] = 0.0          # padding_idx row (model.py:162)
    if tied:
        sd["output.output.weight"] = sd["embedding.weight"]
    return sd


def captions(B, R, T, V, seed, ragged=True, min_len=2):
    """Caption batch in the preprocess.ipynb wire format (cells 15/17):
    [START] + tokens in 1..V-4 + [END] + PAD(0)...; lengths = len(tokens)+1."""
    lengths = integers((B, R), seed, min_len, T) if ragged else np.full((B, R), T - 1, np.int64)
    caps = np.zeros((B, R, T), np.int64)
    toks = integers((B, R, T), seed + 1, 1, V - 3)
    for b in range(B):
        for r in range(R):
            n = int(lengths[b, r])
            caps[b, r, 0] = V - 2
            caps[b, r, 1:n] = toks[b, r, 1:n]
            caps[b, r, n] = V - 1
    return caps, lengths
