"""Test oracle (not shipped): the reference's standard decoding strategies restated in plain Python.

Follows src/transformer/sampling_strategy.cc:29-43 (top_p / top_k cut), :107-147 (SoftMax with temperature), :235-304
(GetSortedTopK through sslib TopKQueue, 3rd_party/sslib/top_k_queue.h:13-22: higher weight first, equal weights: lower id
first), :359-392 (StdSamplingStrategy::ChooseTokens) and 3rd_party/sslib/random.h:15-121 / random.cc:75-146 (Random =
the java.util.Random LCG; RandomSampling with count 1).  Pinned by the published java.util.Random known answers
(seed 42: nextInt() = -1170105035, 234785527; nextDouble() = 0.7275636800328681).
"""
import math

import numpy as np

AUTO, STD, GREEDY, TOP_K, TOP_P = 0, 1, 2, 3, 4
MULT, ADD, MASK = 0x5DEECE66D, 0xB, (1 << 48) - 1


class JavaRandom:
    def __init__(self, seed):
        self.seed = (int(seed) ^ MULT) & MASK

    def next(self, bits):
        self.seed = (self.seed * MULT + ADD) & MASK
        v = self.seed >> (48 - bits)
        return v - (1 << 32) if v >= (1 << 31) else v          # (int32_t) cast

    def next_double(self, lo=0.0, hi=1.0):
        d = ((self.next(26) << 27) + self.next(27)) / float(1 << 53)
        return lo + d * (hi - lo)


def sorted_top_k(logits_f16, k):
    lg = np.asarray(logits_f16, np.float16).astype(np.float32)
    order = sorted(range(lg.size), key=lambda i: (-float(lg[i]), i))
    return [(i, float(lg[i])) for i in order[:k]]


def softmax_pool(pool, temperature):
    if not pool:
        return []
    t = max(np.float32(temperature), np.float32(0.001))
    m = max(np.float32(w) for _, w in pool)
    ws = [np.float32(math.exp(float((np.float32(w) - m) / t))) for _, w in pool]
    s = np.float32(0)
    for w in ws:
        s = np.float32(s + w)
    s = max(s, np.float32(0.00001))
    return [(i, np.float32(w / s)) for (i, _), w in zip(pool, ws)]


def draw_one(rng, pool):
    base, upper = [], 0.0
    for _, w in pool:
        base.append(upper)
        upper += max(0.0, float(w))
    r = rng.next_double(0.0, upper)
    begin, end, mid = 0, len(pool) - 1, 0
    while begin < end:
        mid = (end + begin) // 2
        if r < base[mid]:
            end = mid
        elif r > base[mid + 1]:
            begin = mid + 1
            mid = begin
        else:
            break
    return pool[mid]


def choose_tokens(logits_f16, strategy, rng, max_k=8, top_p=0.9, pool_size=50, temperature=1.0):
    n = np.asarray(logits_f16).size
    qlen = 1 if strategy == GREEDY else min(pool_size, n)
    p = top_p if strategy in (STD, TOP_P) else 1.0
    pool = softmax_pool(sorted_top_k(logits_f16, qlen), temperature)
    top_k = min(len(pool), max_k)
    cut, cum = [], np.float32(0)
    for item in pool:
        cum = np.float32(cum + item[1])
        cut.append(item)
        if cum >= np.float32(p) or len(cut) >= top_k:
            break
    return draw_one(rng, cut), cut
