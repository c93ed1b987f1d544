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
