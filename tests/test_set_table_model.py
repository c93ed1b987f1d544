"""CPU: the scheme csrc/clique.hip (nx_build_fast) uses to lay out a CPython set table WITHOUT inserting the keys one by one, as a Python
model against the live interpreter.

The walk of the reference's tie-break (networkx.find_cliques order) needs the iteration order of sets of small ints built by insertion:
open addressing, slot key & mask, 9 linear probes, then i * 5 + 1 + perturb; tables of 8 / 32 / 128 slots as the set grows, re-inserted in
slot order at each resize (Objects/setobject.c, CPython 3.10).  The device builds the FINAL table in two steps:
  T[s] = the earliest key (insertion index) whose HOME slot is s                     - one parallel pass (ds_min)
  every key that owns its home stays there; the others - "displaced" - are placed one after the other in insertion order: slot s is
  taken when key d arrives iff an earlier key has its home there (T[s] < d) or an earlier displaced key landed there; a LATER home key
  found in the slot taken is displaced in turn.
This file checks that scheme (and the replay of the small tables before it) against `list(set)` of the interpreter that runs the tests,
for the sizes the walk meets (up to 76 members: 128 slots) and a few beyond."""
import random
import sys

import pytest

INF = 1 << 30


def probe_slots(key, mask):
    """the slots set_add_entry looks at for `key`, in order (hash(int) = int for the small non-negative ints used here)"""
    perturb, i = key, key & mask
    while True:
        yield i
        if i + 9 <= mask:
            for j in range(1, 10):
                yield i + j
        perturb >>= 5
        i = (i * 5 + 1 + perturb) & mask


def insert_sequentially(keys, mask):
    """reference model: one key at a time"""
    tab = [None] * (mask + 1)
    for k in keys:
        for s in probe_slots(k, mask):
            if tab[s] is None:
                tab[s] = k
                break
    return tab


def insert_displaced_scheme(keys, mask):
    """the device's scheme for one table: T by home slot, then only the displaced keys, in insertion order"""
    n = len(keys)
    T = [INF] * (mask + 1)
    for d, k in enumerate(keys):
        T[k & mask] = min(T[k & mask], d)
    slot = [k & mask for k in keys]
    pending = sorted(d for d, k in enumerate(keys) if T[k & mask] != d)
    landed = set()
    placed_displaced = 0
    while pending:
        d = pending.pop(0)
        for s in probe_slots(keys[d], mask):
            if T[s] < d or s in landed:
                continue
            if T[s] != INF:                       # its home key comes later: that one is displaced in turn
                assert T[s] > d
                pending.append(T[s]); pending.sort()
            landed.add(s); slot[d] = s
            placed_displaced += 1
            break
    tab = [None] * (mask + 1)
    for d, k in enumerate(keys):
        assert tab[slot[d]] is None
        tab[slot[d]] = k
    return tab, placed_displaced


def grown_set_order(keys, final=insert_displaced_scheme):
    """iteration order of a set built by s.add(k) for k in keys: 5 keys in 8 slots, 19 in 32, 77 in 128, 307 in 512 ... (a resize to
    4 x used once fill * 5 >= mask * 3, keys re-inserted in slot order), the last table by `final`"""
    cuts = [(5, 7), (19, 31), (77, 127), (307, 511), (1229, 2047)]
    n = len(keys)
    size_mask = 7
    for c, m in cuts:
        if n >= c:
            size_mask = {7: 31, 31: 127, 127: 511, 511: 2047, 2047: 8191}[m]
    seq = list(keys)
    for c, m in cuts:
        if n >= c and m < size_mask:
            tab = insert_sequentially(seq[:c], m)
            seq = [k for k in tab if k is not None] + seq[c:]
    tab = final(seq, size_mask)
    tab = tab[0] if isinstance(tab, tuple) else tab
    return [k for k in tab if k is not None]


@pytest.mark.skipif(sys.version_info[:2] != (3, 10), reason="the set implementation restated is CPython 3.10's (the reference's interpreter)")
def test_displaced_key_scheme_gives_the_interpreters_set_order():
    rng = random.Random(5)
    checked = displaced = 0
    for trial in range(6000):
        n = rng.choice([3, 4, 5, 7, 12, 18, 19, 20, 33, 50, 64, 65, 70, 76, 77, 90, 128])
        kmax = rng.choice([40, 96, 139, 178, 256, 300, 512, 1024])
        keys = rng.sample(range(kmax), min(n, kmax))
        s = set()
        for k in keys:
            s.add(k)
        want = list(s)
        assert grown_set_order(keys, insert_sequentially) == want, (trial, keys)           # the replay itself
        got = grown_set_order(keys)
        assert got == want, (trial, keys)
        checked += 1
    # and the final-table scheme alone on tables with many clashes (copies: one table, no growth)
    for trial in range(3000):
        mask = rng.choice([7, 31, 127, 255])
        n = rng.randrange(1, (mask + 1) * 3 // 5 + 1)
        keys = rng.sample(range(4 * (mask + 1)), n)
        a = insert_sequentially(keys, mask)
        b, nd = insert_displaced_scheme(keys, mask)
        assert a == b, (trial, mask, keys)
        displaced += nd
    assert checked == 6000 and displaced > 5000                                             # the test is not vacuous


def test_probe_sequence_matches_setobject_c():
    # LINEAR_PROBES = 9, PERTURB_SHIFT = 5; no linear probes when the window would pass the end of the table
    assert list(zip(range(12), probe_slots(5, 127)))[:11] == list(zip(range(11), [5, 6, 7, 8, 9, 10, 11, 12, 13, 14, (5 * 5 + 1 + (5 >> 5)) & 127]))
    g = probe_slots(125, 127)
    assert next(g) == 125 and next(g) == (125 * 5 + 1 + (125 >> 5)) & 127
    g = probe_slots(3, 7)
    assert [next(g) for _ in range(3)] == [3, (3 * 5 + 1 + 0) & 7, (((3 * 5 + 1) & 7) * 5 + 1 + 0) & 7]
