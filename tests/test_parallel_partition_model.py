"""CPU: the LOGIC of csrc/retrack.hip's wave-parallel k-d tree node build (rb_partition_hoare / rb_nth_element_wave / the scipy
partition pass of rb_build_node_wave), restated in Python next to the sequential algorithms it replaces - libstdc++'s
std::nth_element (introselect: median of three to the front, unguarded Hoare partition, insertion sort below four elements) and
scipy cKDTree's partition pass - on random arrays with heavy ties: the same array, element for element, and the same cut.  (The
first version of the device code took the cut from the list of misplaced positions alone and was wrong when the left pointer runs
into the region the swaps have just filled; tests/test_gpu_full_sequence.py caught it on the device, this test holds the rule.)"""
import random


def _median_to_first(a, first, last, key):
    mid = first + (last - first) // 2
    ia, ib, ic = first + 1, mid, last - 1
    K = lambda i: key(a[i])                                                   # noqa: E731
    if K(ia) < K(ib):
        s = ib if K(ib) < K(ic) else (ic if K(ia) < K(ic) else ia)
    else:
        s = ia if K(ia) < K(ic) else (ic if K(ib) < K(ic) else ib)
    a[first], a[s] = a[s], a[first]


def _insertion(a, first, last, key):
    for i in range(first + 1, last):
        v = a[i]
        if key(v) < key(a[first]):
            for j in range(i, first, -1):
                a[j] = a[j - 1]
            a[first] = v
        else:
            j = i
            while key(v) < key(a[j - 1]):
                a[j] = a[j - 1]
                j -= 1
            a[j] = v


def _hoare_sequential(a, first, last, key):
    f, l, piv = first + 1, last, key(a[first])
    while True:
        while key(a[f]) < piv:
            f += 1
        l -= 1
        while piv < key(a[l]):
            l -= 1
        if not f < l:
            return f
        a[f], a[l] = a[l], a[f]
        f += 1


def _hoare_parallel(a, first, last, key):
    piv = key(a[first])
    L = [p for p in range(first + 1, last) if key(a[p]) >= piv]               # ascending
    R = [p for p in range(first, last) if key(a[p]) <= piv]                   # read from its end: descending
    nL, nR = len(L), len(R)
    kk = 0
    while kk < min(nL, nR) and L[kk] < R[nR - 1 - kk]:
        kk += 1
    for i in range(kk):                                                       # independent swaps
        x, y = L[i], R[nR - 1 - i]
        a[x], a[y] = a[y], a[x]
    return min(L[kk] if kk < nL else last, R[nR - kk] if kk > 0 else last)


def _nth(a, first, nth, last, key, partition):
    a = list(a)
    depth = 2 * ((last - first).bit_length() - 1)
    while last - first > 3:
        if depth == 0:
            return None                                                       # (introselect's heap fallback: sequential on the device too)
        depth -= 1
        _median_to_first(a, first, last, key)
        cut = partition(a, first, last, key)
        if cut <= nth:
            first = cut
        else:
            last = cut
    _insertion(a, first, last, key)
    return a


def _scipy_sequential(a, start, end, split, key):
    a, p, q = list(a), start, end - 1
    while p <= q:
        if key(a[p]) < split:
            p += 1
        elif key(a[q]) >= split:
            q -= 1
        else:
            a[p], a[q] = a[q], a[p]
            p += 1
            q -= 1
    return a, p


def _scipy_parallel(a, start, end, split, key):
    a = list(a)
    L = [p for p in range(start, end) if key(a[p]) >= split]
    R = [p for p in range(start, end) if key(a[p]) < split]
    nL, nR = len(L), len(R)
    kk = 0
    while kk < min(nL, nR) and L[kk] < R[nR - 1 - kk]:
        kk += 1
    for i in range(kk):
        x, y = L[i], R[nR - 1 - i]
        a[x], a[y] = a[y], a[x]
    return a, start + nR


def test_parallel_partitions_equal_the_sequential_algorithms():
    rng = random.Random(1)
    key = lambda e: e[0]                                                      # noqa: E731
    done = 0
    for _ in range(4000):
        n = rng.randint(101, 700)
        span = rng.choice([3, 10, 50, 2000])                                  # 3: nearly everything ties
        a = [(rng.randint(0, span), i) for i in range(n)]
        s = rng.randint(0, 5)
        nth = s + (n - s) // 2
        A = _nth(a, s, nth, n, key, _hoare_sequential)
        B = _nth(a, s, nth, n, key, _hoare_parallel)
        if A is None or B is None:
            assert A is None and B is None
            continue
        assert A == B, (n, span)
        split = key(A[nth])
        assert _scipy_sequential(A, s, n, split, key) == _scipy_parallel(A, s, n, split, key), (n, span)
        done += 1
    assert done > 3900


# ---------------------------------------------------------------------------------------------- npy_aquicksort, segment by segment
def aquicksort_par(v):
    """segments in any order; partition by the list rule; insertion sorts at the end"""
    n = len(v); ts = list(range(n))
    if n < 2: return ts
    cdepth = 0; k = n
    while k > 1: k >>= 1; cdepth += 1
    cdepth *= 2
    work = [(0, n - 1, cdepth, True)]; leaves = []
    while work:
        pl, pr, cd, pushed = work.pop(rng_order.randrange(len(work)))       # any order
        if pushed and cd < 0: raise RuntimeError("heapsort")
        if pr - pl <= 16: leaves.append((pl, pr)); continue
        pm = pl + ((pr - pl) >> 1)
        if v[ts[pm]] < v[ts[pl]]: ts[pm], ts[pl] = ts[pl], ts[pm]
        if v[ts[pr]] < v[ts[pm]]: ts[pr], ts[pm] = ts[pm], ts[pr]
        if v[ts[pm]] < v[ts[pl]]: ts[pm], ts[pl] = ts[pl], ts[pm]
        vp = v[ts[pm]]
        ts[pm], ts[pr - 1] = ts[pr - 1], ts[pm]
        L = [p for p in range(pl + 1, pr) if not (v[ts[p]] < vp)]            # ascending; ends with pr - 1
        R = [p for p in range(pr - 2, pl - 1, -1) if not (vp < v[ts[p]])]     # descending; ends with pl
        kk = 0
        while kk < min(len(L), len(R)) and L[kk] < R[kk]: kk += 1
        for i in range(kk): ts[L[i]], ts[R[i]] = ts[R[i]], ts[L[i]]
        pi = L[0] if kk == 0 else min(L[kk], R[kk - 1])
        ts[pi], ts[pr - 1] = ts[pr - 1], ts[pi]
        cd -= 1
        if pi - pl < pr - pi:
            work.append((pi + 1, pr, cd, True)); work.append((pl, pi - 1, cd, False))
        else:
            work.append((pl, pi - 1, cd, True)); work.append((pi + 1, pr, cd, False))
    for pl, pr in leaves:
        for i in range(pl + 1, pr + 1):
            vi = ts[i]; j = i
            while j > pl and v[vi] < v[ts[j - 1]]: ts[j] = ts[j - 1]; j -= 1
            ts[j] = vi
    return ts



def test_aquicksort_segments_in_any_order_with_list_rule_partitions():
    """csrc/retrack.hip rb_aquicksort_wave: NumPy 1.22's argsort (npy_aquicksort) with its segments processed in ANY order - each with
    the depth budget of its parent minus one - and every Hoare loop replaced by the two ordered lists of stop positions, against the
    oracle's restatement of the sequential algorithm (itself pinned against NumPy 1.22.3 outputs): two-valued keys (the detector's
    sigmas), all-equal keys, few and many distinct keys, 1 to 1 500 elements"""
    import numpy as np
    import oracle
    global rng_order
    rng_order = random.Random(9)
    rng = random.Random(3)
    for t in range(1200):
        n = rng.choice([1, 2, 17, 18, 40, 100, 257, 530, 1024, 1500])
        nv = rng.choice([1, 2, 2, 2, 3, 10, 1000])
        v = [float(rng.randrange(nv)) for _ in range(n)]
        if nv == 2:
            f = rng.random()
            v = [1.0 if rng.random() < f else 2.0 for _ in range(n)]
        want = oracle.argsort_numpy122(np.array(v)).tolist()
        assert aquicksort_par(v) == want, (t, n, nv)
