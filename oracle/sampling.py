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


# ---- MinP / TFS / Typical / Mirostat (sampling_strategy.cc:696-760, :787-876, :901-990, :1015-1098)
MIN_P, TFS, TYPICAL, MIROSTAT = 7, 8, 9, 10
F = np.float32


def cut_min_p(pool, min_p=0.05):
    out = [pool[0]]
    for it in pool[1:]:
        if F(it[1]) < F(F(min_p) * F(pool[0][1])):
            break
        out.append(it)
    return out


def cut_tfs(pool, z=0.95):
    out = [pool[0]]
    if len(pool) < 3:
        return out
    w = [F(p) for _, p in pool]
    d1 = [F(w[i] - w[i + 1]) for i in range(len(w) - 1)]
    d2 = [F(abs(F(d1[i] - d1[i + 1]))) for i in range(len(d1) - 1)]
    s = F(0)
    for v in d2:
        s = F(s + v)
    d2 = [F(v / s) for v in d2] if s > F(1e-6) else [F(F(1.0) / F(len(d2))) for _ in d2]
    cum = F(0)
    for i in range(1, len(d2)):
        cum = F(cum + d2[i])
        if cum > F(z):
            break
        out.append(pool[i])
    return out


def cut_typical(pool, p=0.95):
    w = [F(x) for _, x in pool]
    ent = F(0)
    for x in w:
        ent = F(ent + F(-x * F(np.log(x, dtype=np.float32))))
    shifted = [F(abs(F(F(-np.log(x, dtype=np.float32)) - ent))) for x in w]
    idx = sorted(range(len(pool)), key=lambda i: (float(shifted[i]), i))
    out = [pool[idx[0]]]
    cum = F(0)
    for i in idx[1:]:
        cum = F(cum + w[i])
        if cum > F(p):
            break
        out.append(pool[i])
    return out


def choose_tokens_ex(logits_f16, strategy, rng, temperature=1.0, pool_size=50, min_p=0.05, z=0.95, typical_p=0.95, eta=0.1, tau=5.0, mu=None):
    """Returns ((token, prob), cut pool, mu after the draw)."""
    n = np.asarray(logits_f16).size
    raw = sorted_top_k(logits_f16, min(pool_size, n))
    pool = softmax_pool(raw, temperature)
    if strategy == MIN_P:
        cut = cut_min_p(pool, min_p)
    elif strategy == TFS:
        cut = cut_tfs(pool, z)
    elif strategy == TYPICAL:
        cut = cut_typical(pool, typical_p)
    else:
        mu = F(2.0 * tau) if mu is None else F(mu)
        k = 0
        while k < len(pool) and not (F(-np.log2(F(pool[k][1]), dtype=np.float32)) > mu):
            k += 1
        cut = softmax_pool(raw[:max(k, 1)], temperature)
    sel = draw_one(rng, cut)
    if strategy == MIROSTAT:
        mu = F(mu - F(eta) * F(F(-np.log2(F(sel[1]), dtype=np.float32)) - F(tau)))
    return sel, cut, mu


# ---- FSD / RandomizedFSD (sampling_strategy.cc:457-541, :569-667; NGram: sampling_strategy.h:125-236)
FSD, RANDOM_FSD = 5, 6


class NGram:
    def __init__(self, n=3, beta=0.9):
        self.n, self.beta, self.tokens, self.following = n, F(beta), [], None

    def initialize(self, tokens):
        self.tokens = list(tokens)
        self.following = [dict() for _ in range(self.n)]
        for order in range(1, self.n + 1):
            for i in range(0, len(tokens) - order + 1):
                gram = tuple(tokens[i:i + order])
                self.following[order - 1].setdefault(gram[:-1], []).append(gram[-1])

    def update(self, tok):
        if self.following is None:
            self.following = [dict() for _ in range(self.n)]
        for i in range(self.n):
            if len(self.tokens) < i:
                continue
            key = tuple(self.tokens[len(self.tokens) - i:]) if i else ()
            self.following[i].setdefault(key, []).append(tok)
        self.tokens.append(tok)

    def penalize(self, cands):
        pen = {}
        if len(self.tokens) < self.n - 1 or self.following is None:
            return pen
        for c in cands:
            remaining, score = F(1), F(0)
            for i in range(self.n - 1, -1, -1):
                key = tuple(self.tokens[len(self.tokens) - i:]) if i else ()
                nxt = self.following[i].get(key, [])
                cnt = sum(1 for v in nxt if v == c)
                if cnt == 0:
                    continue
                if i == 0:
                    score = F(score + F(remaining * F(F(cnt) / F(len(nxt)))))
                else:
                    score = F(score + F(F(remaining * self.beta) * F(F(cnt) / F(len(nxt) + 1))))
                remaining = F(remaining - F(remaining * self.beta))
            pen[c] = score
        return pen


def _top_k_of_pool(pool, k):
    return sorted(pool, key=lambda it: (-float(it[1]), it[0]))[:k]


class FsdState:
    def __init__(self, n=3, beta=0.9):
        self.ngram, self.started, self.new_tokens = NGram(n, beta), False, 0


def _fsd_pick(raw, temperature, st, text, k, alpha):
    pool = _top_k_of_pool(softmax_pool(raw, temperature), k)
    if not st.started:
        st.ngram.initialize(text)
        st.started = True
    pen = st.ngram.penalize([i for i, _ in pool])
    pool = [(i, F(F(F(1) - F(alpha)) * F(w) - F(F(alpha) * pen[i])) if i in pen else w) for i, w in pool]
    cut = _top_k_of_pool(pool, len(pool))
    return cut[0], cut


def choose_tokens_fsd(logits_f16, strategy, rng, st, text, temperature=1.0, pool_size=50, k=6, alpha=0.5, max_k=8, top_p=0.93):
    n = np.asarray(logits_f16).size
    raw = sorted_top_k(logits_f16, min(pool_size, n))
    if strategy == FSD or st.new_tokens >= 10 or F(F(rng.next(24)) / F(1 << 24)) >= F(0.5):
        sel, cut = _fsd_pick(raw, temperature, st, text, k, alpha)
    else:
        pool = softmax_pool(raw, temperature)
        cut, cum = [], F(0)
        for item in pool:
            cum = F(cum + item[1])
            cut.append(item)
            if cum >= F(top_p) or len(cut) >= min(len(pool), max_k):
                break
        sel = draw_one(rng, cut)
    st.ngram.update(sel[0])
    st.new_tokens += 1
    return sel, cut
