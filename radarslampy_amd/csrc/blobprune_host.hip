// Host entry points of the order-dependent blob bookkeeping (blobprune.h): pure CPU code inside libroam_hip.so, used by
// radarslampy_amd/getFeatures.py for the stage-level blob_doh / adaptiveNMS mirrors.  The engine's device-side retrack
// (retrack.hip) runs the same functions on the GPU.
#include "roam_internal.h"
#include <math.h>
#include <vector>
#include "blobprune.h"

extern "C" {

// skimage.feature.blob._prune_blobs (reference getFeatures.py:47-51 via blob_doh), pair order included.
// blobs (n,3) f64 rows [row, col, sigma] in peak_local_max order; rows / cols integer valued in [0, 32767].
// keep_out (n) u8 = 1 for the surviving blobs.  No GPU involved.
int32_t roam_prune_blobs(const double *blobs, int32_t n, double overlap, uint8_t *keep_out)
{
    if (n < 0 || (n > 0 && (!blobs || !keep_out))) return ROAM_E_ARG;
    if (n == 0) return ROAM_OK;
    if (n > 32767) return ROAM_E_CAPACITY;
    std::vector<int16_t> xy(2 * (size_t)n), idx(n);
    double smax = blobs[2];
    for (int i = 0; i < n; i++) {
        const double r = blobs[3 * i], c = blobs[3 * i + 1];
        if (!(r >= 0 && r <= 32767 && c >= 0 && c <= 32767) || r != floor(r) || c != floor(c)) return ROAM_E_ARG;
        xy[2 * i] = (int16_t)r; xy[2 * i + 1] = (int16_t)c;
        smax = blobs[3 * i + 2] > smax ? blobs[3 * i + 2] : smax;
    }
    const double distance = 2 * smax * sqrt(2.0);
    const int node_cap = 2 * n + 8;
    std::vector<BpNode> nodes(node_cap);
    int stack[3 * 64];
    const int nn = bp_build(xy.data(), n, idx.data(), nodes.data(), node_cap, stack);
    if (nn < 0) return ROAM_E_CAPACITY;
    const int task_cap = 64 * nn + 64;
    std::vector<BpTask> tasks(task_cap);
    std::vector<int> st(3 * 1024);
    BpTracker tr;
    const int nt = bp_tasks(xy.data(), n, nodes.data(), distance, tasks.data(), task_cap, st.data(), 1024, tr);
    if (nt < 0) return ROAM_E_CAPACITY;
    std::vector<uint32_t> pairs(BP_MAX_PAIRS);
    const int np = bp_expand(xy.data(), idx.data(), nodes.data(), tasks.data(), nt, tr.ub, pairs.data(), BP_MAX_PAIRS);
    if (np < 0) return ROAM_E_CAPACITY;
    std::vector<uint16_t> tabA(131072), tabB(131072);
    std::vector<uint16_t> order(np > 0 ? np : 1);
    if (bp_pyset_order(pairs.data(), np, tabA.data(), 131072, tabB.data(), 131072, order.data()) != np) return ROAM_E_CAPACITY;
    std::vector<double> sig(n);
    for (int i = 0; i < n; i++) sig[i] = blobs[3 * i + 2];
    for (int k = 0; k < np; k++) {
        const uint32_t pr = pairs[order[k]];
        const int i = (int)(pr >> 16), j = (int)(pr & 0xffffu);
        if (sig[i] == 0 || sig[j] == 0) continue;           // a pruned member: the reference's pass changes nothing
        if (bp_overlaps(blobs[3 * i], blobs[3 * i + 1], sig[i], blobs[3 * j], blobs[3 * j + 1], sig[j], overlap)) {
            if (sig[i] > sig[j]) sig[j] = 0; else sig[i] = 0;
        }
    }
    for (int i = 0; i < n; i++) keep_out[i] = sig[i] > 0;
    return ROAM_OK;
}

// np.argsort(keys) with the tie order of the reference's pinned NumPy 1.22.3 (adaptiveNMS, getFeatures.py:69)
int32_t roam_argsort_np122(const double *keys, int32_t n, int32_t *order_out)
{
    if (n < 0 || (n > 0 && (!keys || !order_out))) return ROAM_E_ARG;
    bp_aquicksort(keys, n, order_out);
    return ROAM_OK;
}

}  // extern "C"
