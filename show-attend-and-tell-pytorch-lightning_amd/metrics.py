"""Validation metrics of ``SAT.score_captions`` (reference model.py:646-682) without nltk.

The reference calls ``nltk.translate.bleu_score.corpus_bleu`` and ``nltk.translate.gleu_score.corpus_gleu``
(nltk 3.6.2, requirements.txt:6 -- a third-party dependency that is not installed here).  The two functions below restate
the published algorithms of those nltk functions over token-id lists:

* BLEU (Papineni et al. 2002) as nltk computes it at corpus level: clipped n-gram counts and hypothesis n-gram totals summed
  over the corpus per order, brevity penalty from the summed closest reference lengths, geometric mean over the weights
  with nltk's default ``SmoothingFunction().method0`` (an order without any match contributes ``log(sys.float_info.min)``),
  0 when there is no unigram match.
* GLEU (Wu et al. 2016) at corpus level: per hypothesis the reference with the best tp / max(tp+fp, tp+fn) over all 1..4-grams,
  matches and totals summed over the corpus.

Parity: unpinned against nltk itself (absent).  Pinned against the reference's own independent BLEU, ``token_bleu``
(dev/dev_corpus_metrics.py:19-55, imported to generate tests/golden/g10_metrics.npz) wherever the two definitions coincide
(every weighted order has a match), and against hand-computed cases."""
import math
import sys
from collections import Counter
from fractions import Fraction


def _ngrams(seq, n):
    return [tuple(seq[i:i + n]) for i in range(len(seq) - n + 1)]


def modified_precision(references, hypothesis, n):
    """clipped matches / hypothesis n-grams of one segment (nltk.translate.bleu_score.modified_precision)"""
    counts = Counter(_ngrams(hypothesis, n)) if len(hypothesis) >= n else Counter()
    max_counts = {}
    for ref in references:
        ref_counts = Counter(_ngrams(ref, n)) if len(ref) >= n else Counter()
        for ng in counts:
            max_counts[ng] = max(max_counts.get(ng, 0), ref_counts[ng])
    clipped = sum(min(c, max_counts.get(ng, 0)) for ng, c in counts.items())
    return clipped, max(1, sum(counts.values()))


def closest_ref_length(references, hyp_len):
    return min((len(r) for r in references), key=lambda rl: (abs(rl - hyp_len), rl))


def brevity_penalty(closest_ref_len, hyp_len):
    if hyp_len > closest_ref_len:
        return 1.0
    if hyp_len == 0:
        return 0.0
    return math.exp(1 - closest_ref_len / hyp_len)


def corpus_bleu(list_of_references, hypotheses, weights=(0.25, 0.25, 0.25, 0.25)):
    assert len(list_of_references) == len(hypotheses), "one reference set per hypothesis"
    num, den = Counter(), Counter()
    hyp_lengths = ref_lengths = 0
    for references, hypothesis in zip(list_of_references, hypotheses):
        for i in range(1, len(weights) + 1):
            c, t = modified_precision(references, hypothesis, i)
            num[i] += c; den[i] += t
        hyp_lengths += len(hypothesis)
        ref_lengths += closest_ref_length(references, len(hypothesis))
    bp = brevity_penalty(ref_lengths, hyp_lengths)
    if num[1] == 0:
        return 0
    total = 0.0
    terms = []
    for i, w in enumerate(weights, start=1):
        p = Fraction(num[i], den[i]) if num[i] != 0 else None          # method0: no match at this order -> the smallest float
        terms.append(w * math.log(p if p is not None else sys.float_info.min))
    total = math.fsum(terms)
    return bp * math.exp(total)


def corpus_gleu(list_of_references, hypotheses, min_len=1, max_len=4):
    assert len(list_of_references) == len(hypotheses), "one reference set per hypothesis"

    def everygrams(seq):
        return Counter(ng for n in range(min_len, max_len + 1) for ng in _ngrams(seq, n))

    n_match = n_all = 0
    for references, hypothesis in zip(list_of_references, hypotheses):
        hyp = everygrams(hypothesis)
        tpfp = sum(hyp.values())
        best = None
        for ref in references:
            rg = everygrams(ref)
            tpfn = sum(rg.values())
            tp = sum((rg & hyp).values())
            total = max(tpfp, tpfn)
            if total > 0 and (best is None or tp / total > best[0] / best[1]):
                best = (tp, total)
        if best is not None:
            n_match += best[0]; n_all += best[1]
    return 0.0 if n_all == 0 else n_match / n_all
