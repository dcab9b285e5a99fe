"""Test oracle (not shipped): the reference's standard decoding strategies restated in plain Python.

Follows src/transformer/sampling_strategy.cc:29-43 (top_p / top_k cut), :107-147 (SoftMax with temperature), :235-304
(GetSortedTopK through sslib TopKQueue, 3rd_party/sslib/top_k_queue.h:13-22: higher weight first, equal weights: lower id
first), :359-392 (StdSamplingStrategy::ChooseTokens) and 3rd_party/sslib/random.h:15-121 / random.cc:75-146 (Random =
the java.util.Random LCG; RandomSampling with count 1).  PINNED to the reference itself: tests/golden/ref_sampling.npz holds
the draws of the reference's own sampling_strategy.cc (compiled where it lies: oracle/Makefile `ref_sampling`, driver
oracle/ref_sampling_driver.cc, generator tests/golden/gen_sampling_fixtures.py) for all ten strategies, and
tests/test_sampling_ref_fixtures.py checks this file AND the product's host sampler against them draw for draw -- which is
how the std::sort tie order below and Mirostat's second SoftMax over the sorted prefix were found.  Also pinned by the
published java.util.Random known answers (seed 42: nextInt() = -1170105035, 234785527; nextDouble() = 0.7275636800328681).
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


def std_sort(a, less):
    """std::sort of libstdc++ (GCC's bits/stl_algo.h, the library the reference is built with here): introsort -- median-of-3
    quicksort down to runs of 16, heapsort past a depth of 2 * floor(log2 n), one final insertion sort.  NOT stable: the
    reference sorts its token pool with it (sampling_strategy.cc:113, :935), so which of several EQUAL logits comes first
    is this algorithm's doing.  Restated from the published algorithm; sorts the list in place."""
    n = len(a)
    if n < 2:
        return a

    def swap(i, j):
        a[i], a[j] = a[j], a[i]

    def move_median_to_first(result, x, y, z):
        if less(a[x], a[y]):
            if less(a[y], a[z]):
                swap(result, y)
            elif less(a[x], a[z]):
                swap(result, z)
            else:
                swap(result, x)
        elif less(a[x], a[z]):
            swap(result, x)
        elif less(a[y], a[z]):
            swap(result, z)
        else:
            swap(result, y)

    def unguarded_partition(first, last, pivot):
        while True:
            while less(a[first], a[pivot]):
                first += 1
            last -= 1
            while less(a[pivot], a[last]):
                last -= 1
            if not first < last:
                return first
            swap(first, last)
            first += 1

    def heap_sort(first, last):                      # std::partial_sort(first, last, last): make_heap + sort_heap
        def adjust(hole, length, val):
            top = hole
            child = hole
            while child < (length - 1) // 2:
                child = 2 * (child + 1)
                if less(a[first + child], a[first + child - 1]):
                    child -= 1
                a[first + hole] = a[first + child]
                hole = child
            if (length & 1) == 0 and child == (length - 2) // 2:
                child = 2 * (child + 1)
                a[first + hole] = a[first + child - 1]
                hole = child - 1
            parent = (hole - 1) // 2
            while hole > top and less(a[first + parent], val):
                a[first + hole] = a[first + parent]
                hole = parent
                parent = (hole - 1) // 2
            a[first + hole] = val
        length = last - first
        if length >= 2:
            parent = (length - 2) // 2
            while True:
                adjust(parent, length, a[first + parent])
                if parent == 0:
                    break
                parent -= 1
        while last - first > 1:
            last -= 1
            val = a[last]
            a[last] = a[first]
            adjust(0, last - first, val)

    def introsort_loop(first, last, depth):
        while last - first > 16:
            if depth == 0:
                heap_sort(first, last)
                return
            depth -= 1
            mid = first + (last - first) // 2
            move_median_to_first(first, first + 1, mid, last - 1)
            cut = unguarded_partition(first + 1, last, first)
            introsort_loop(cut, last, depth)
            last = cut

    def unguarded_linear_insert(i):
        val = a[i]
        nxt = i - 1
        while less(val, a[nxt]):
            a[i] = a[nxt]
            i = nxt
            nxt -= 1
        a[i] = val

    def insertion_sort(first, last):
        for i in range(first + 1, last):
            if less(a[i], a[first]):
                val = a[i]
                a[first + 1:i + 1] = a[first:i]
                a[first] = val
            else:
                unguarded_linear_insert(i)

    introsort_loop(0, n, 2 * (n.bit_length() - 1))
    if n > 16:
        insertion_sort(0, 16)
        for i in range(16, n):
            unguarded_linear_insert(i)
    else:
        insertion_sort(0, n)
    return a


def softmax_pool(pool, temperature):
    if not pool:
        return []
    pool = std_sort(list(pool), lambda x, y: np.float32(x[1]) > np.float32(y[1]))       # SoftMax sorts first (sampling_strategy.cc:113)
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
    idx = std_sort(list(range(len(pool))), lambda x, y: shifted[x] < shifted[y])                # std::sort, sampling_strategy.cc:935
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
        rawl = dict(raw)                                   # the first k entries of the SORTED pool with their logits, softmaxed (and sorted) again
        cut = softmax_pool([(i, rawl[i]) for i, _ in pool[:max(k, 1)]], temperature)
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
