// Sparse Cholesky for the sparse branch of kkt_chol2 (reference src/python/misc.py:1405-1462, :1528-1558):
// the device replacement of cholmod.symbolic / cholmod.numeric / cholmod.solve (reference src/C/cholmod.c:273-558).
// CHOLMOD itself (SuiteSparse, third party, not vendored) is not restated; what is reproduced is the contract
// "analyse S once, refactor numerically every iteration, solve with the factor" and the unique result L L' = P S P'.
//
//   host, once per factory (symbolic_analyze):
//       pattern of S = H + G'D^2G (lower, CSC)  ->  nested-dissection ordering (BFS level-structure separators,
//       George 1973)  ->  elimination tree, postorder, column structures  ->  fundamental supernodes (+ relaxed
//       amalgamation of small chains)  ->  per supernode: columns, row list, children, extend-add index maps,
//       levels of the supernodal tree, and the gather lists that turn (H values, G values, di) into S values.
//   device, every factor():  assemble_S_kernel (S values straight into the supernodal panels) and one
//       front_kernel launch per tree level (multifrontal: one workgroup per frontal matrix: extend-add of the
//       children's update matrices, blocked dense partial Cholesky, update matrix left for the parent).
//   device, every solve():   level-by-level forward (leaves -> root) and backward (root -> leaves) supernodal
//       substitution; children's right-hand-side updates are extend-added by the parent (deterministic, no atomics).
#include <algorithm>
#include <numeric>
#include <queue>

#include <cstdlib>
#include <cstring>

#include "kkt_common.h"
#include "ordering.h"
#include <chrono>
#include <memory>
#include <future>
#include <atomic>

namespace mi355kkt {

// =====================================================================================================
// host: symbolic analysis
// =====================================================================================================
namespace {

// lower-triangular pattern of S = H + G'G (values irrelevant) as adjacency lists of the full symmetric graph
void build_graph(int n, int m, const int64_t* gcp, const int64_t* gri, const int64_t* hcp, const int64_t* hri,
                 std::vector<std::vector<int>>& adj) {
    adj.assign(n, {});
    // rows of G -> cliques
    std::vector<std::vector<int>> rows(m);
    for (int j = 0; j < n; ++j)
        for (int64_t k = gcp[j]; k < gcp[j + 1]; ++k) rows[gri[k]].push_back(j);
    for (int r = 0; r < m; ++r) {
        const auto& c = rows[r];
        for (size_t a = 0; a < c.size(); ++a)
            for (size_t b = 0; b < c.size(); ++b)
                if (a != b) adj[c[a]].push_back(c[b]);
    }
    if (hcp)
        for (int j = 0; j < n; ++j)
            for (int64_t k = hcp[j]; k < hcp[j + 1]; ++k) {
                const int i = (int)hri[k];
                if (i > j) {   // only the lower triangle of H is meaningful
                    adj[i].push_back(j);
                    adj[j].push_back(i);
                }
            }
    for (auto& a : adj) {
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
    }
}

}  // namespace

int sp_wide_threshold();

int symbolic_analyze(SparseSymbolic& S, int n, int m, const int64_t* gcp, const int64_t* gri, const int64_t* hcp,
                     const int64_t* hri) {
    S = SparseSymbolic();
    S.n = n;
    S.m = m;
    if (n == 0) return 0;
    const bool dbg = dev_knob("MI355KKT_SPARSE_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sparse] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<std::vector<int>> adj;
    build_graph(n, m, gcp, gri, hcp, hri, adj);
    lap("pattern of S");
    // ---- ordering
    std::vector<int> order;
    OrderingInfo oinfo;
    fill_reducing_ordering(adj, order, 0, &oinfo);
    if ((int)order.size() != n) {
        set_last_error("symbolic_analyze: ordering has %d entries, expected %d", (int)order.size(), n);
        return -1;
    }
    {
        std::vector<char> seen(n, 0);
        for (int k = 0; k < n; ++k) {
            if (order[k] < 0 || order[k] >= n || seen[order[k]]) {
                set_last_error("symbolic_analyze: ordering is not a permutation (position %d)", k);
                return -1;
            }
            seen[order[k]] = 1;
        }
    }
    S.order_method = oinfo.method;
    if (dbg)
        fprintf(stderr, "[sparse] ordering: %s (nested dissection nnz %lld flops %.3e height %d | minimum degree nnz %lld flops %.3e height %d)\n",
                oinfo.method == 2 ? "approximate minimum degree" : "nested dissection", (long long)oinfo.nnz_nd, oinfo.flops_nd,
                oinfo.levels_nd, (long long)oinfo.nnz_amd, oinfo.flops_amd, oinfo.levels_amd);
    lap("ordering");
    S.perm = order;                 // perm[new] = old
    S.iperm.assign(n, 0);
    for (int k = 0; k < n; ++k) S.iperm[order[k]] = k;
    // ---- permuted lower pattern (CSC): column j holds rows i > j
    std::vector<std::vector<int>> low(n);
    for (int v = 0; v < n; ++v) {
        const int j = S.iperm[v];
        for (int u : adj[v]) {
            const int i = S.iperm[u];
            if (i > j) low[j].push_back(i);
        }
    }
    for (auto& c : low) std::sort(c.begin(), c.end());
    // ---- elimination tree and column counts cc[j] = |struct(j)| (rows > j of L(:, j)) of the permuted matrix: computed
    //      by the ordering step for its cost model (ordering.cpp, etree_and_counts), already in postorder
    const std::vector<int>& parent = oinfo.parent;
    const std::vector<int64_t>& cc = oinfo.colcount;
    if ((int)parent.size() != n || (int)cc.size() != n) {
        set_last_error("symbolic_analyze: ordering returned no elimination tree");
        return -1;
    }
    // ---- supernodes (relaxed amalgamation of chains, ordering.cpp)
    std::vector<int> sn_first;   // first column of each supernode
    std::vector<int> sn_of;
    relaxed_supernodes(parent, cc, sn_first, sn_of);
    const int ns = (int)sn_first.size();
    S.ns = ns;
    S.sn_first = sn_first;
    S.sn_first.push_back(n);
    // rows of supernode s: its own columns, then the union of its columns' structures outside the supernode, by a
    // supernodal symbolic factorisation: own pattern entries below the supernode U below-rows of the child supernodes
    // (a supernode's parent is the supernode of its first below-row, so children are complete before their parent)
    S.sn_rowptr.assign(ns + 1, 0);
    std::vector<std::vector<int>> snrows(ns);
    {
        std::vector<int> flag(n, -1);
        std::vector<std::vector<int>> kids(ns);
        for (int s = 0; s < ns; ++s) {
            const int f = S.sn_first[s], l = S.sn_first[s + 1];
            std::vector<int>& r = snrows[s];
            for (int j = f; j < l; ++j) r.push_back(j);
            std::vector<int> below;
            for (int j = f; j < l; ++j)
                for (int i : low[j])
                    if (i >= l && flag[i] != s) { flag[i] = s; below.push_back(i); }
            for (int c : kids[s]) {
                const std::vector<int>& rc = snrows[c];
                const int wc = S.sn_first[c + 1] - S.sn_first[c];
                for (size_t k = wc; k < rc.size(); ++k) {
                    const int i = rc[k];
                    if (i >= l && flag[i] != s) { flag[i] = s; below.push_back(i); }
                }
            }
            std::sort(below.begin(), below.end());
            r.insert(r.end(), below.begin(), below.end());
            S.sn_rowptr[s + 1] = S.sn_rowptr[s] + (int64_t)r.size();
            if (!below.empty()) kids[sn_of[below[0]]].push_back(s);
        }
    }
    S.sn_rows.resize(S.sn_rowptr[ns]);
    for (int s = 0; s < ns; ++s) std::copy(snrows[s].begin(), snrows[s].end(), S.sn_rows.begin() + S.sn_rowptr[s]);
    lap("supernodes + row lists");
    // supernodal tree: parent = supernode of the first row below the supernode
    S.sn_parent.assign(ns, -1);
    std::vector<std::vector<int>> snchild(ns);
    for (int s = 0; s < ns; ++s) {
        const int w = S.sn_first[s + 1] - S.sn_first[s];
        if ((int)snrows[s].size() > w) {
            S.sn_parent[s] = sn_of[snrows[s][w]];
            snchild[S.sn_parent[s]].push_back(s);
        }
    }
    // levels (height from the leaves), panel / update-matrix offsets
    S.sn_level.assign(ns, 0);
    int maxlevel = 0;
    for (int s = 0; s < ns; ++s) {   // children have smaller indices
        for (int c : snchild[s]) S.sn_level[s] = std::max(S.sn_level[s], S.sn_level[c] + 1);
        maxlevel = std::max(maxlevel, S.sn_level[s]);
    }
    S.nlevels = maxlevel + 1;
    S.level_ptr.assign(S.nlevels + 1, 0);
    for (int s = 0; s < ns; ++s) S.level_ptr[S.sn_level[s] + 1]++;
    for (int l = 0; l < S.nlevels; ++l) S.level_ptr[l + 1] += S.level_ptr[l];
    S.level_sn.resize(ns);
    {
        std::vector<int> pos(S.level_ptr.begin(), S.level_ptr.end() - 1);
        for (int s = 0; s < ns; ++s) S.level_sn[pos[S.sn_level[s]]++] = s;
    }
    // storage: one buffer.  Small front: panel (h x w, ld h) followed by its update matrix (hu x hu, ld hu).
    // Big front (handled by the dense MFMA kernels): the whole h x h frontal matrix, panel = its first w columns,
    // update matrix = its trailing block (ld h).
    S.wide_threshold = sp_wide_threshold();     // latched here for the life of this analysis
    S.panel_off.assign(ns + 1, 0);
    S.upd_off.assign(ns, 0);
    S.upd_ld.assign(ns, 0);
    S.big.assign(ns, 0);
    int64_t off = 0;
    const double big_flops = dev_knob("MI355KKT_SPARSE_BIG_FLOPS") ? atof(dev_knob("MI355KKT_SPARSE_BIG_FLOPS")) : 5.0e4;
    const int64_t big_h = dev_knob("MI355KKT_SPARSE_BIG_H") ? atoi(dev_knob("MI355KKT_SPARSE_BIG_H")) : 48;
    for (int s = 0; s < ns; ++s) {
        const int64_t h = S.sn_rowptr[s + 1] - S.sn_rowptr[s], w = S.sn_first[s + 1] - S.sn_first[s];
        const double fl = (double)w * h * h;
        S.flops += fl;   // rough
        S.nnzL += h * w;
        S.panel_off[s] = off;
        // (a wide supernode is always a big front: its solves use the extend-add boundary tables that only big fronts' children carry)
        if ((fl >= big_flops && h >= big_h) || w > S.wide_threshold) {
            S.big[s] = 1;
            S.upd_off[s] = off + w + w * h;
            S.upd_ld[s] = (int)h;
            off += h * h;
        } else {
            S.upd_off[s] = off + h * w;
            S.upd_ld[s] = (int)(h - w);
            off += h * w + (h - w) * (h - w);
        }
        off = (off + 1) & ~(int64_t)1;   // keep 16-byte alignment of every front
    }
    S.panel_off[ns] = off;
    S.store_doubles = off;
    // inside a level: small fronts first (one batched launch), big fronts after them (dense kernels, one by one)
    S.level_nsmall.assign(S.nlevels, 0);
    for (int l = 0; l < S.nlevels; ++l) {
        std::stable_partition(S.level_sn.begin() + S.level_ptr[l], S.level_sn.begin() + S.level_ptr[l + 1],
                              [&](int sn) { return !S.big[sn]; });
        for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) S.level_nsmall[l] += !S.big[S.level_sn[k]];
    }
    // big fronts of every level as one descriptor list for the level-batched dense kernels
    S.vb.clear();
    S.vb_ptr.assign(S.nlevels + 1, 0);
    S.vb_maxh.assign(S.nlevels, 0);
    S.vb_maxw.assign(S.nlevels, 0);
    S.vb_maxcount = 0;
    for (int l = 0; l < S.nlevels; ++l) {
        for (int k = S.level_ptr[l] + S.level_nsmall[l]; k < S.level_ptr[l + 1]; ++k) {
            const int sn = S.level_sn[k];
            const int h = (int)(S.sn_rowptr[sn + 1] - S.sn_rowptr[sn]), w = S.sn_first[sn + 1] - S.sn_first[sn];
            S.vb.push_back(VbDesc{S.panel_off[sn], h, w, S.sn_first[sn], sn});
            S.vb_maxh[l] = std::max(S.vb_maxh[l], h);
            S.vb_maxw[l] = std::max(S.vb_maxw[l], w);
        }
        S.vb_ptr[l + 1] = (int)S.vb.size();
        S.vb_maxcount = std::max(S.vb_maxcount, S.vb_ptr[l + 1] - S.vb_ptr[l]);
    }
    // work list of the persistent tile kernel (potrf.hip: potrf_tiles_vb_kernel), per level: tiles (i >= j) of every big front
    // with the tile boundaries 0, 128, .., 128 q, w, w + 128, ..; order (j, front, i) so that every front's chain starts at once
    S.tv_tickets.clear();
    S.tv_ptr.assign(S.nlevels + 1, 0);
    S.tv_prog_off.assign(S.vb.size(), 0);
    S.tv_linv_off.assign(S.vb.size(), 0);
    S.tv_nprog.assign(S.nlevels, 0);
    S.tv_prog_max = S.tv_linv_max = 0;
    for (int l = 0; l < S.nlevels; ++l) {
        int po = 0, lo = 0, maxnt = 0;
        std::vector<int> ntt(S.vb_ptr[l + 1] - S.vb_ptr[l]);
        for (int f = S.vb_ptr[l]; f < S.vb_ptr[l + 1]; ++f) {
            const VbDesc& dd = S.vb[f];
            const int q = dd.w / 128, wr = dd.w - q * 128, ntf = q + (wr > 0 ? 1 : 0);
            const int nt = ntf + (dd.h - dd.w + 127) / 128;
            ntt[f - S.vb_ptr[l]] = nt;
            S.tv_prog_off[f] = po;
            S.tv_linv_off[f] = lo;
            po += nt;
            lo += ntf;
            maxnt = std::max(maxnt, nt);
        }
        for (int j = 0; j < maxnt; ++j)
            for (int f = S.vb_ptr[l]; f < S.vb_ptr[l + 1]; ++f) {
                const int nt = ntt[f - S.vb_ptr[l]];
                for (int i = j; i < nt; ++i) {
                    S.tv_tickets.push_back(f - S.vb_ptr[l]);
                    S.tv_tickets.push_back(i);
                    S.tv_tickets.push_back(j);
                    S.tv_tickets.push_back(0);
                }
            }
        S.tv_ptr[l + 1] = (int)(S.tv_tickets.size() / 4);
        S.tv_nprog[l] = po;
        S.tv_prog_max = std::max(S.tv_prog_max, po);
        S.tv_linv_max = std::max(S.tv_linv_max, lo);
    }
    // supernodes whose off-diagonal panel is large: their solve-phase products run as multi-workgroup kernels
    S.heavy_ptr.assign(S.nlevels + 1, 0);
    S.heavy_maxhu.assign(S.nlevels, 0);
    S.heavy_maxw.assign(S.nlevels, 0);
    S.heavy.clear();
    for (int l = 0; l < S.nlevels; ++l) {
        for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
            const int sn = S.level_sn[k];
            const int64_t h = S.sn_rowptr[sn + 1] - S.sn_rowptr[sn], w = S.sn_first[sn + 1] - S.sn_first[sn];
            if ((h - w) * w > 32768 || (w > S.wide_threshold && h > w)) {      // == SP_HEAVY / SP_WIDE in the kernels
                S.heavy.push_back(sn);
                S.heavy_maxhu[l] = std::max<int>(S.heavy_maxhu[l], (int)(h - w));
                S.heavy_maxw[l] = std::max<int>(S.heavy_maxw[l], (int)w);
            }
        }
        S.heavy_ptr[l + 1] = (int)S.heavy.size();
    }
    // supernodes wider than one workgroup's LDS vector (256): their diagonal blocks are solved by the persistent dense
    // triangular solve (blas2.hip), launched one by one (they are the few top separators of the tree)
    S.wide.clear();
    S.wide_ptr.assign(S.nlevels + 1, 0);
    for (int l = 0; l < S.nlevels; ++l) {
        for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k)
            if (S.sn_first[S.level_sn[k] + 1] - S.sn_first[S.level_sn[k]] > S.wide_threshold) S.wide.push_back(S.level_sn[k]);
        S.wide_ptr[l + 1] = (int)S.wide.size();
    }
    if (dev_knob("MI355KKT_SPARSE_DEBUG")) {
        fprintf(stderr, "[sparse] n=%d supernodes=%d levels=%d store=%.1f MB\n", n, ns, S.nlevels, off * 8.0 / 1e6);
        for (int l = 0; l < S.nlevels; ++l) {
            int nb = 0, nsm = 0, maxh_s = 0, maxh_b = 0, maxw_b = 0;
            double fs = 0, fb = 0, fmax_s = 0;
            for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
                const int sn = S.level_sn[k];
                const int64_t h = S.sn_rowptr[sn + 1] - S.sn_rowptr[sn], w = S.sn_first[sn + 1] - S.sn_first[sn];
                const double fl = (double)w * h * h;
                if (S.big[sn]) { nb++; fb += fl; maxh_b = std::max<int>(maxh_b, (int)h); maxw_b = std::max<int>(maxw_b, (int)w); }
                else { nsm++; fs += fl; fmax_s = std::max(fmax_s, fl); maxh_s = std::max<int>(maxh_s, (int)h); }
            }
            fprintf(stderr, "[sparse] level %3d: small %6d (flops %.2e, max %.2e, max h %4d) big %4d (flops %.2e, max h %5d, max w %3d)\n",
                    l, nsm, fs, fmax_s, maxh_s, nb, fb, maxh_b, maxw_b);
        }
    }
    // children lists + extend-add maps: position of each below-row of child c inside the parent's row list
    S.child_ptr.assign(ns + 1, 0);
    for (int s = 0; s < ns; ++s) S.child_ptr[s + 1] = S.child_ptr[s] + (int)snchild[s].size();
    S.child_list.resize(S.child_ptr[ns]);
    S.relmap_off.assign(ns + 1, 0);
    for (int s = 0; s < ns; ++s) {
        std::copy(snchild[s].begin(), snchild[s].end(), S.child_list.begin() + S.child_ptr[s]);
        const int64_t h = S.sn_rowptr[s + 1] - S.sn_rowptr[s], w = S.sn_first[s + 1] - S.sn_first[s];
        S.relmap_off[s + 1] = S.relmap_off[s] + (h - w);
    }
    S.relmap.resize(S.relmap_off[ns]);
    {
        std::vector<int> where(n, -1);
        for (int p = 0; p < ns; ++p) {
            for (size_t k = 0; k < snrows[p].size(); ++k) where[snrows[p][k]] = (int)k;
            for (int c : snchild[p]) {
                const int w = S.sn_first[c + 1] - S.sn_first[c];
                for (size_t k = w; k < snrows[c].size(); ++k) {
                    const int pos = where[snrows[c][k]];
                    if (pos < 0) {
                        set_last_error("symbolic_analyze: child row missing in parent front");
                        return -1;
                    }
                    S.relmap[S.relmap_off[c] + (k - w)] = pos;
                }
            }
            for (int r : snrows[p]) where[r] = -1;
        }
    }
    // extend-add of the big fronts works on 64-column x 256-row tiles of the parent: for every child of a big front the
    // first child row that lands at or beyond each multiple of 64 of the parent's row positions (so a tile finds its share
    // of a child with four table reads instead of four binary searches through global memory)
    S.ea_off.assign(ns + 1, 0);
    for (int p = 0; p < ns; ++p)
        for (int c : snchild[p]) {
            const int64_t hp = S.sn_rowptr[p + 1] - S.sn_rowptr[p];
            S.ea_off[c + 1] = S.big[p] ? (hp + 63) / 64 + 1 : 0;
        }
    for (int s2 = 0; s2 < ns; ++s2) S.ea_off[s2 + 1] += S.ea_off[s2];
    S.ea_lb.assign(S.ea_off[ns], 0);
    for (int c = 0; c < ns; ++c) {
        const int64_t nb = S.ea_off[c + 1] - S.ea_off[c];
        if (nb == 0) continue;
        const int* rm = S.relmap.data() + S.relmap_off[c];
        const int hc = (int)(S.relmap_off[c + 1] - S.relmap_off[c]);
        int k = 0;
        for (int64_t t = 0; t < nb; ++t) {
            while (k < hc && rm[k] < t * 64) ++k;
            S.ea_lb[S.ea_off[c] + t] = k;
        }
    }
    lap("tree, levels, maps");
    // ---- numeric assembly lists: every structural nonzero of S in (permuted) column j, row i >= j, gets a slot in
    //      its supernode's panel; contributions: H entries and products G_ra G_rb di_r^2.
    //      Sorted by target slot (CSR over targets) so that one thread sums one entry in a fixed order.
    struct Contrib { int64_t slot; int a, b, r; };   // a = G nz index (or H nz index with b = -1)
    std::vector<Contrib> cs;
    auto slot_of = [&](int inew, int jnew) -> int64_t {   // i >= j (permuted)
        const int s = sn_of[jnew];
        const std::vector<int>& r = snrows[s];
        const int64_t h = (int64_t)r.size();
        const int col = jnew - S.sn_first[s];
        const auto it = std::lower_bound(r.begin() + (inew >= S.sn_first[s + 1] ? (S.sn_first[s + 1] - S.sn_first[s]) : 0),
                                         r.end(), inew);
        int pos;
        if (inew < S.sn_first[s + 1]) pos = inew - S.sn_first[s];
        else {
            if (it == r.end() || *it != inew) return -1;
            pos = (int)(it - r.begin());
        }
        return S.panel_off[s] + (int64_t)col * h + pos;
    };
    {
        // G by rows (CSR: column, nz index), columns ascending inside a row like the column-major input
        const int64_t gnz = gcp[n];
        std::vector<int64_t> rp((size_t)m + 1, 0);
        for (int64_t k = 0; k < gnz; ++k) rp[gri[k] + 1]++;
        for (int r = 0; r < m; ++r) rp[r + 1] += rp[r];
        std::vector<int> rcol((size_t)gnz), rnz((size_t)gnz);
        {
            std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
            for (int j = 0; j < n; ++j)
                for (int64_t k = gcp[j]; k < gcp[j + 1]; ++k) {
                    const int64_t q = fill[gri[k]]++;
                    rcol[q] = j;
                    rnz[q] = (int)k;
                }
        }
        size_t total = 0;
        for (int r = 0; r < m; ++r) total += (size_t)((rp[r + 1] - rp[r]) * (rp[r + 1] - rp[r] + 1) / 2);
        cs.reserve(total + (hcp ? (size_t)hcp[n] : 0));
        for (int r = 0; r < m; ++r)
            for (int64_t qa = rp[r]; qa < rp[r + 1]; ++qa)
                for (int64_t qb = rp[r]; qb < rp[r + 1]; ++qb) {
                    const int ia = S.iperm[rcol[qa]], ib = S.iperm[rcol[qb]];
                    if (ia < ib) continue;
                    const int64_t sl = slot_of(ia, ib);
                    if (sl < 0) { set_last_error("symbolic_analyze: G'G entry outside the symbolic pattern"); return -1; }
                    cs.push_back({sl, rnz[qa], rnz[qb], r});
                }
        if (hcp)
            for (int j = 0; j < n; ++j)
                for (int64_t k = hcp[j]; k < hcp[j + 1]; ++k) {
                    const int i = (int)hri[k];
                    if (i < j) continue;   // lower triangle only
                    int ia = S.iperm[i], ib = S.iperm[j];
                    if (ia < ib) std::swap(ia, ib);
                    const int64_t sl = slot_of(ia, ib);
                    if (sl < 0) { set_last_error("symbolic_analyze: H entry outside the symbolic pattern"); return -1; }
                    cs.push_back({sl, (int)k, -1, 0});
                }
    }
    lap("assembly contributions");
    {   // by target slot, ties in generation order (the summation order of sp_assemble_kernel is part of the plan): a sort
        // of (slot, sequence number) keys -- unique, so no stable sort of the 24-byte records -- in four chunks on their own
        // threads, merged pairwise, then one gather
        const size_t N = cs.size();
        std::vector<std::pair<int64_t, uint32_t>> key(N);
        for (size_t k = 0; k < N; ++k) key[k] = {cs[k].slot, (uint32_t)k};
        if (N >= ((size_t)1 << 32)) { set_last_error("symbolic_analyze: more than 2^32 assembly contributions"); return -1; }
        const size_t q1 = N / 4, q2 = N / 2, q3 = N - N / 4;
        auto srt = [&](size_t a, size_t b) { std::sort(key.begin() + a, key.begin() + b); };
        if (N > 100000) {
            std::future<void> f1, f2, f3;
            try {
                f1 = std::async(std::launch::async, srt, (size_t)0, q1);
                f2 = std::async(std::launch::async, srt, q1, q2);
                f3 = std::async(std::launch::async, srt, q2, q3);
            } catch (const std::system_error&) {
            }
            srt(q3, N);
            if (f1.valid()) f1.get(); else srt(0, q1);
            if (f2.valid()) f2.get(); else srt(q1, q2);
            if (f3.valid()) f3.get(); else srt(q2, q3);
            std::future<void> m1;
            try {
                m1 = std::async(std::launch::async, [&]() { std::inplace_merge(key.begin(), key.begin() + q1, key.begin() + q2); });
            } catch (const std::system_error&) {
            }
            std::inplace_merge(key.begin() + q2, key.begin() + q3, key.end());
            if (m1.valid()) m1.get(); else std::inplace_merge(key.begin(), key.begin() + q1, key.begin() + q2);
            std::inplace_merge(key.begin(), key.begin() + q2, key.end());
        } else {
            srt(0, N);
        }
        std::vector<Contrib> sorted(N);
        for (size_t k = 0; k < N; ++k) sorted[k] = cs[key[k].second];
        cs.swap(sorted);
    }
    for (size_t k = 0; k < cs.size();) {
        size_t e = k;
        while (e < cs.size() && cs[e].slot == cs[k].slot) ++e;
        S.asm_slot.push_back(cs[k].slot);
        S.asm_ptr.push_back((int64_t)k);
        k = e;
    }
    lap("assembly lists (build + sort)");
    S.asm_ptr.push_back((int64_t)cs.size());
    S.asm_a.resize(cs.size());
    S.asm_b.resize(cs.size());
    S.asm_r.resize(cs.size());
    for (size_t k = 0; k < cs.size(); ++k) {
        S.asm_a[k] = cs[k].a;
        S.asm_b[k] = cs[k].b;
        S.asm_r[k] = cs[k].r;
    }
    return 0;
}

// =====================================================================================================
// device: numeric factorisation and solves
// =====================================================================================================
__global__ __launch_bounds__(256) void sp_assemble_kernel(int64_t ntargets, const int64_t* __restrict__ slot,
                                                          const int64_t* __restrict__ ptr, const int* __restrict__ a,
                                                          const int* __restrict__ b, const int* __restrict__ r,
                                                          const double* __restrict__ gv, const double* __restrict__ hv,
                                                          const double* __restrict__ di, double* __restrict__ panels) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ntargets) return;
    double s = 0.0;
    for (int64_t k = ptr[t]; k < ptr[t + 1]; ++k) {
        if (b[k] < 0) s += hv[a[k]];
        else {
            const double d = di[r[k]];
            s += (d * gv[a[k]]) * (d * gv[b[k]]);      // (di G)_ra (di G)_rb, as the reference forms Gs first
        }
    }
    panels[slot[t]] = s;
}

struct SpDev {   // device copies of the symbolic structure (plain pointers for the kernels)
    const int* sn_first;
    const int64_t* sn_rowptr;
    const int* sn_rows;
    const int64_t* panel_off;
    const int64_t* upd_off;
    const int* upd_ld;
    const int* child_ptr;
    const int* child_list;
    const int64_t* relmap_off;
    const int* relmap;
    const int* level_sn;
    const int64_t* ea_off;     // per child of a big front: offset of its tile-boundary table in ea_lb
    const int* ea_lb;
    int wide;                  // sp_wide_threshold()
};

// clears the panels (the h x w columns to be factored) of all supernodes: chunk c = `len[c]` doubles at store + off[c] (at most
// SP_ZERO_CHUNK each; the list is built once, sparse_engine_create).  The update matrices need no clearing pass: the small fronts
// clear their own (sp_front_kernel), the big fronts' are cleared tile by tile inside sp_extend_add_vb_kernel.
constexpr int SP_ZERO_CHUNK = 32768;
__global__ __launch_bounds__(256) void sp_zero_chunks_kernel(const int64_t* __restrict__ off, const int* __restrict__ len,
                                                             double* __restrict__ store) {
    double* __restrict__ p = store + off[blockIdx.x];
    const int n = len[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += 256) p[i] = 0.0;
}

// One workgroup = one frontal matrix.  F = [ L-panel (h x w) | U (h-w x h-w) ]: the panel already holds the
// entries of S; U starts as the extend-add of the children's update matrices.
__global__ __launch_bounds__(256) void sp_front_kernel(SpDev d, int level_begin, double* panels, double* upd,
                                                       int* __restrict__ info) {
    const int s = d.level_sn[level_begin + blockIdx.x];
    const int tid = threadIdx.x;
    const int w = d.sn_first[s + 1] - d.sn_first[s];
    const int h = (int)(d.sn_rowptr[s + 1] - d.sn_rowptr[s]);
    const int hu = h - w;
    double* __restrict__ P = panels + d.panel_off[s];   // h x w, column-major
    double* __restrict__ U = upd + d.upd_off[s];        // hu x hu (ld = hu for these small fronts), lower part used
    __shared__ double piv;
    __shared__ int bad;
    if (tid == 0) bad = 0;
    // ---- U := 0, then extend-add the children (their update matrices and the part that lands in the panel)
    for (int64_t e = tid; e < (int64_t)hu * hu; e += 256) U[e] = 0.0;
    __syncthreads();
    for (int ci = d.child_ptr[s]; ci < d.child_ptr[s + 1]; ++ci) {
        const int c = d.child_list[ci];
        const int wc = d.sn_first[c + 1] - d.sn_first[c];
        const int hc = (int)(d.sn_rowptr[c + 1] - d.sn_rowptr[c]) - wc;
        const double* __restrict__ Uc = upd + d.upd_off[c];
        const int ldc = d.upd_ld[c];
        const int* __restrict__ rm = d.relmap + d.relmap_off[c];
        for (int64_t e = tid; e < (int64_t)hc * hc; e += 256) {
            const int i = (int)(e % hc), j = (int)(e / hc);
            if (i < j) continue;
            const int pi = rm[i], pj = rm[j];            // positions in this front; pi >= pj (row lists are sorted)
            const double v = Uc[i + (int64_t)j * ldc];
            if (pj < w) P[pi + (int64_t)pj * h] += v;
            else U[(pi - w) + (int64_t)(pj - w) * hu] += v;
        }
        __syncthreads();   // children are added one after the other: deterministic, no atomics
    }
    // ---- blocked partial Cholesky of the first w columns (block size 16)
    for (int jb = 0; jb < w; jb += 16) {
        const int pw = min(16, w - jb);
        for (int jj = 0; jj < pw; ++jj) {
            const int j = jb + jj;
            if (tid == 0) {
                const double a = P[j + (int64_t)j * h];
                if (!(a > 0.0)) { if (!bad) bad = d.sn_first[s] + j + 1; piv = 1.0; }
                else piv = sqrt(a);
            }
            __syncthreads();
            const double dj = piv, inv = 1.0 / piv;
            for (int i = j + tid; i < h; i += 256) P[i + (int64_t)j * h] = (i == j) ? dj : P[i + (int64_t)j * h] * inv;
            __syncthreads();
            // rank-1 update restricted to the remaining columns of this block
            const int nc = jb + pw - 1 - j;
            for (int64_t e = tid; e < (int64_t)nc * (h - j - 1); e += 256) {
                const int c = j + 1 + (int)(e / (h - j - 1)), i = j + 1 + (int)(e % (h - j - 1));
                if (i >= c) P[i + (int64_t)c * h] -= P[i + (int64_t)j * h] * P[c + (int64_t)j * h];
            }
            __syncthreads();
        }
        // rank-pw update of the columns to the right inside the panel ...
        const int c0 = jb + pw;
        for (int64_t e = tid; e < (int64_t)(w - c0) * (h - c0); e += 256) {
            const int c = c0 + (int)(e / (h - c0)), i = c0 + (int)(e % (h - c0));
            if (i < c) continue;
            double sacc = 0.0;
            for (int k = jb; k < jb + pw; ++k) sacc += P[i + (int64_t)k * h] * P[c + (int64_t)k * h];
            P[i + (int64_t)c * h] -= sacc;
        }
        // ... and of the update matrix
        for (int64_t e = tid; e < (int64_t)hu * hu; e += 256) {
            const int i = (int)(e % hu), c = (int)(e / hu);
            if (i < c) continue;
            double sacc = 0.0;
            for (int k = jb; k < jb + pw; ++k) sacc += P[(w + i) + (int64_t)k * h] * P[(w + c) + (int64_t)k * h];
            U[i + (int64_t)c * hu] -= sacc;
        }
        __syncthreads();
    }
    if (tid == 0 && bad) atomicMin(info, bad);      // smallest failing column (info starts at INT_MAX)
}

// extend-add of the children's update matrices into the big fronts of one level: grid (64-column blocks, 256-row
// blocks, fronts).  One workgroup owns its block of the target front, so the children can be added one after the
// other (deterministic, no atomics); the child rows / columns that land in the block are contiguous ranges of its
// sorted map.
constexpr int EA_COLS = 64, EA_ROWS = 64;
__global__ __launch_bounds__(256) void sp_extend_add_vb_kernel(SpDev d, const VbDesc* __restrict__ vb, double* store) {
    const VbDesc dd = vb[blockIdx.z];
    const int s = dd.pad;                      // supernode id
    const int h = dd.h;
    const int c0 = blockIdx.x * EA_COLS;
    const int r0 = blockIdx.y * EA_ROWS, r1 = min(r0 + EA_ROWS, h);
    if (c0 >= h || r0 >= h || r1 <= c0) return;            // outside the front / strictly above the diagonal
    double* __restrict__ F = store + dd.off;
    const int nbt = (h + 63) / 64;             // last entry of a child's boundary table (= its number of update rows)
    {   // round 4: the front's Schur part (rows and columns beyond its w panel columns) starts from zero HERE, tile by tile, right
        // before the children are added (the lines stay in the L2) -- sparse_engine_factor clears only the panels (sp_zero_chunks_kernel)
        // instead of the whole store, whose big fronts are h x h squares with an unused upper triangle: 3.9 GB at 64^3
        const int wp = dd.w;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int i = r0 + lane;
        if (i < r1 && i >= wp)
            for (int j = max(c0, wp) + wave; j < min(c0 + EA_COLS, h); j += 4) F[i + (int64_t)j * h] = 0.0;
        __syncthreads();
    }
    for (int ci = d.child_ptr[s]; ci < d.child_ptr[s + 1]; ++ci) {
        const int c = d.child_list[ci];
        const int hc = (int)(d.sn_rowptr[c + 1] - d.sn_rowptr[c]) - (d.sn_first[c + 1] - d.sn_first[c]);
        if (hc <= 0) continue;
        const int* __restrict__ rm = d.relmap + d.relmap_off[c];
        // first child index whose parent position is >= c0, c1, r0, r1 (tile boundaries are multiples of 64)
        const int* __restrict__ lb = d.ea_lb + d.ea_off[c];
        const int ja = lb[blockIdx.x], jb = lb[min((int)blockIdx.x + 1, nbt)];
        const int ia = max(lb[min((int)blockIdx.y, nbt)], ja), ib = lb[min((int)blockIdx.y + 1, nbt)];
        if (jb > ja && ib > ia) {
            const double* __restrict__ Uc = store + d.upd_off[c];
            const int ldc = d.upd_ld[c];
            // a lane owns a row of the child's update matrix inside the tile (its target row is looked up once), the four
            // waves split the tile's columns: consecutive lanes read consecutive entries of a column, short dependent chains
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int nj = jb - ja;
            const int jq0 = ja + (nj * wave) / 4, jq1 = ja + (nj * (wave + 1)) / 4;
            for (int i = ia + lane; i < ib; i += 64) {
                const int ri = rm[i];
                const double* __restrict__ u = Uc + i;
                const int jend = min(jq1, i + 1);         // lower triangle: i >= j
                int j = jq0;
                for (; j + 8 <= jend; j += 8) {           // eight independent read-modify-writes in flight (distinct columns)
                    int64_t at[8];
                    double fv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) at[q] = ri + (int64_t)rm[j + q] * h;
#pragma unroll
                    for (int q = 0; q < 8; ++q) fv[q] = F[at[q]];
#pragma unroll
                    for (int q = 0; q < 8; ++q) fv[q] += u[(int64_t)(j + q) * ldc];
#pragma unroll
                    for (int q = 0; q < 8; ++q) F[at[q]] = fv[q];
                }
                for (; j < jend; ++j) F[ri + (int64_t)rm[j] * h] += u[(int64_t)j * ldc];
            }
        }
        __syncthreads();
    }
}

__global__ void sp_merge_info_vb_kernel(const int* __restrict__ local, int n, int* __restrict__ global) {
    const int z = blockIdx.x * 256 + threadIdx.x;
    if (z < n && local[z] > 0) atomicMin(global, local[z]);
}

// (a negative local word is a hand-off timeout of the dense tile kernel: it wins over every pivot index and reaches the caller)
__global__ void sp_merge_info_kernel(const int* __restrict__ local, int offset, int* __restrict__ global) {
    if (*local > 0) atomicMin(global, offset + *local);
    else if (*local < 0) atomicMin(global, *local);
}

// ---- triangular solves with one supernode's w x w diagonal block (w <= 256), right-hand side in LDS ----------------
// 32-column blocks: wave 0 solves the 32 x 32 diagonal block with its rows (columns for the transposed solve) in
// registers and v_readlane-style broadcasts -- no barrier on the 32-step dependency chain -- then all four waves apply
// the block to the rest of the vector.
constexpr int SPB = 32;
constexpr int64_t SP_HEAVY = 32768;     // supernodes with more off-diagonal panel entries get the multi-workgroup kernels
// supernodes wider than this leave the one-workgroup solve kernels: gather / dense persistent trsv / multi-workgroup products
// (128; at most 256, the capacity of the kernels' LDS vector; $MI355KKT_SP_WIDE: experiments.  46^3: solve 1.98 ms at 256,
// 1.76 ms at 128, 1.74 ms at 64)
int sp_wide_threshold() {
    const int v = dev_knob("MI355KKT_SP_WIDE") ? std::min(256, std::max(32, atoi(dev_knob("MI355KKT_SP_WIDE")))) : 128;
    return v;
}

__device__ __forceinline__ double sp_bcast(double v, int srclane) {   // srclane wave-uniform: two v_readlane, no LDS round trip
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void sp_trsv_fwd_lds(const double* __restrict__ P, int h, int w, double* xs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int jb = 0; jb < w; jb += SPB) {
        const int nbk = min(SPB, w - jb);
        if (wave == 0) {
            // round 4: the chain is multiply + broadcast + FMA per column (the reciprocal of the lane's pivot is formed once, off the
            // chain; a division per step was 12-15 dependent instructions) and ends at the block's width (a one-column supernode
            // used to walk all 32 steps)
            double Lr[SPB];
            const int r = min(lane, nbk - 1);
#pragma unroll
            for (int k = 0; k < SPB; ++k) Lr[k] = (k < nbk) ? P[(jb + r) + (int64_t)(jb + k) * h] : 0.0;
            const double dinv = 1.0 / P[(jb + r) + (int64_t)(jb + r) * h];
            double xi = (lane < nbk) ? xs[jb + lane] : 0.0;
#pragma unroll
            for (int k = 0; k < SPB; ++k) {
                if (k < nbk) {                                   // wave-uniform
                    const double v = sp_bcast(xi * dinv, k);      // x_k (lane k's value is final here)
                    if (lane == k) xi = v;
                    if (lane > k) xi = fma(-Lr[k], v, xi);
                }
            }
            if (lane < nbk) xs[jb + lane] = xi;
        }
        __syncthreads();
        for (int i = jb + nbk + tid; i < w; i += 256) {
            const double* __restrict__ Pi = P + i + (int64_t)jb * h;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int k = 0;
            for (; k + 8 <= nbk; k += 8) {           // eight loads in flight per thread (the loop is latency-bound: round 4;
                double v[8];                         //  all 32 of a block at once cost registers, i.e. leaf-level occupancy)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = Pi[(int64_t)(k + q) * h];
                a0 += v[0] * xs[jb + k];
                a1 += v[1] * xs[jb + k + 1];
                a2 += v[2] * xs[jb + k + 2];
                a3 += v[3] * xs[jb + k + 3];
                a0 += v[4] * xs[jb + k + 4];
                a1 += v[5] * xs[jb + k + 5];
                a2 += v[6] * xs[jb + k + 6];
                a3 += v[7] * xs[jb + k + 7];
            }
            for (; k < nbk; ++k) a0 += Pi[(int64_t)k * h] * xs[jb + k];
            xs[i] -= (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void sp_trsv_bwd_lds(const double* __restrict__ P, int h, int w, double* xs) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = (w + SPB - 1) / SPB;
    for (int b = nblk - 1; b >= 0; --b) {
        const int jb = b * SPB;
        const int nbk = min(SPB, w - jb);
        if (wave == 0) {
            double Lc[SPB];      // Lc[k] = L[jb + k][jb + lane]  (column `lane` of the block)
            const int c = min(lane, nbk - 1);
#pragma unroll
            for (int k = 0; k < SPB; ++k) Lc[k] = (k < nbk) ? P[(jb + k) + (int64_t)(jb + c) * h] : 0.0;
            const double dinv = 1.0 / P[(jb + c) + (int64_t)(jb + c) * h];
            double xi = (lane < nbk) ? xs[jb + lane] : 0.0;
#pragma unroll
            for (int k = SPB - 1; k >= 0; --k) {
                if (k < nbk) {                                   // wave-uniform
                    const double v = sp_bcast(xi * dinv, k);
                    if (lane == k) xi = v;
                    if (lane < k) xi = fma(-Lc[k], v, xi);
                }
            }
            if (lane < nbk) xs[jb + lane] = xi;
        }
        __syncthreads();
        {   // columns to the left: xs[i] -= sum_k L[jb + k][i] xs[jb + k]; 32 lanes share one column (contiguous in memory)
            const int k = tid & 31, g = tid >> 5;
            const double xk = (k < nbk) ? xs[jb + k] : 0.0;
            const double* __restrict__ Pk = P + jb + min(k, nbk - 1);
            int i = g;
            for (; i + 24 < jb; i += 32) {            // four columns (loads) in flight per thread
                double v0 = Pk[(int64_t)i * h] * xk, v1 = Pk[(int64_t)(i + 8) * h] * xk;
                double v2 = Pk[(int64_t)(i + 16) * h] * xk, v3 = Pk[(int64_t)(i + 24) * h] * xk;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    v0 += __shfl_xor(v0, o, 64);
                    v1 += __shfl_xor(v1, o, 64);
                    v2 += __shfl_xor(v2, o, 64);
                    v3 += __shfl_xor(v3, o, 64);
                }
                if (k == 0) { xs[i] -= v0; xs[i + 8] -= v1; xs[i + 16] -= v2; xs[i + 24] -= v3; }
            }
            for (; i < jb; i += 8) {
                double v = Pk[(int64_t)i * h] * xk;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (k == 0) xs[i] -= v;
            }
        }
        __syncthreads();
    }
}

// ---- small supernodes (round 4): ONE WAVE per supernode, four per workgroup, no workgroup barrier ---------------------------------
// At 64^3 the lowest four levels hold 33 000 of the 37 000 supernodes, none wider than 22 columns or taller than 144 rows (level 0:
// 20 893 supernodes of 1.7 columns on average): a 256-thread workgroup with two barriers per 32 columns for each of them is
// latency and occupancy, not work (level 0: 82 us forward, 104 us backward).  Supernodes with w <= SP_SMALL_W columns and
// hu <= SP_SMALL_HU rows below the diagonal block take these kernels -- vector and remainder in the wave's slice of LDS, cross-lane
// traffic through LDS / readlane, children one after the other (deterministic) -- the others sp_fwd_kernel / sp_bwd_kernel, which
// skip what is small.
constexpr int SP_SMALL_W = 32, SP_SMALL_HU = 192;
__device__ __forceinline__ bool sp_is_small(int w, int hu) { return w <= SP_SMALL_W && hu <= SP_SMALL_HU; }

// one wave: the whole forward step of the small supernode s (f, w, h, hu: its first column, width, height, rows below);
// `xs`: SP_SMALL_W + SP_SMALL_HU doubles of LDS owned by this wave
__device__ __forceinline__ void sp_fwd_small_body(const SpDev& d, int s, int f, int w, int h, int hu, int lane,
                                                  const double* __restrict__ panels, double* __restrict__ x,
                                                  double* __restrict__ rem, const int64_t* __restrict__ rem_off, double* xs) {
    double* R = xs + SP_SMALL_W;
    const double* __restrict__ P = panels + d.panel_off[s];
    // operands of the diagonal block first (independent of the children): lane = row r of L11
    const int r = min(lane, w - 1);
    double Lr[SP_SMALL_W];
#pragma unroll
    for (int c = 0; c < SP_SMALL_W; ++c) Lr[c] = (c < w) ? P[r + (int64_t)c * h] : 0.0;
    const double dinv = 1.0 / P[r + (int64_t)r * h];
    if (lane < w) xs[lane] = x[f + lane];
    for (int i = lane; i < hu; i += 64) R[i] = 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int ci = d.child_ptr[s]; ci < d.child_ptr[s + 1]; ++ci) {
        const int c = d.child_list[ci];
        const int hc = (int)(d.sn_rowptr[c + 1] - d.sn_rowptr[c]) - (d.sn_first[c + 1] - d.sn_first[c]);
        const double* __restrict__ Rc = rem + rem_off[c];
        const int* __restrict__ rm = d.relmap + d.relmap_off[c];
        for (int i = lane; i < hc; i += 64) {            // a child's rows land on distinct positions: plain read-modify-writes
            const int p = rm[i];
            double* dst = (p < w) ? xs + p : R + (p - w);
            *dst += Rc[i];
        }
        __builtin_amdgcn_wave_barrier();
    }
    double xi = (lane < w) ? xs[lane] : 0.0;
#pragma unroll
    for (int c = 0; c < SP_SMALL_W; ++c) {
        if (c < w) {                                     // wave-uniform
            const double v = sp_bcast(xi * dinv, c);
            if (lane == c) xi = v;
            if (lane > c) xi = fma(-Lr[c], v, xi);
        }
    }
    if (lane < w) {
        xs[lane] = xi;
        x[f + lane] = xi;
    }
    __builtin_amdgcn_wave_barrier();
    double* __restrict__ Rg = rem + rem_off[s];
    for (int i = lane; i < hu; i += 64) {
        const double* __restrict__ Pi = P + w + i;
        double a0 = 0.0, a1 = 0.0;
        int j = 0;
        for (; j + 8 <= w; j += 8) {                     // eight loads in flight
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = Pi[(int64_t)(j + q) * h];
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                a0 += v[q] * xs[j + q];
                a1 += v[q + 1] * xs[j + q + 1];
            }
        }
        for (; j + 2 <= w; j += 2) {
            a0 += Pi[(int64_t)j * h] * xs[j];
            a1 += Pi[(int64_t)(j + 1) * h] * xs[j + 1];
        }
        if (j < w) a0 += Pi[(int64_t)j * h] * xs[j];
        Rg[i] = R[i] - (a0 + a1);
    }
}

__global__ __launch_bounds__(256) void sp_fwd_small_kernel(SpDev d, int level_begin, int count, const double* __restrict__ panels,
                                                           double* __restrict__ x, double* __restrict__ rem,
                                                           const int64_t* __restrict__ rem_off, int64_t xstride, int64_t remstride) {
    __shared__ double lds[4][SP_SMALL_W + SP_SMALL_HU];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + wave;
    if (k >= count) return;
    x += (int64_t)blockIdx.y * xstride;
    rem += (int64_t)blockIdx.y * remstride;
    const int s = d.level_sn[level_begin + k];
    const int f = d.sn_first[s];
    const int w = d.sn_first[s + 1] - f;
    const int h = (int)(d.sn_rowptr[s + 1] - d.sn_rowptr[s]);
    sp_fwd_small_body(d, s, f, w, h, h - w, lane, panels, x, rem, rem_off, lds[wave]);
}

__device__ __forceinline__ void sp_bwd_small_body(const SpDev& d, int s, int f, int w, int h, int hu, int lane,
                                                  const double* __restrict__ panels, double* __restrict__ x) {
    const double* __restrict__ P = panels + d.panel_off[s];
    const int* __restrict__ rows = d.sn_rows + d.sn_rowptr[s] + w;
    // lane = column c of L11 for the transposed solve: Lc[k] = L[k][c]
    const int c = min(lane, w - 1);
    double Lc[SP_SMALL_W];
#pragma unroll
    for (int q = 0; q < SP_SMALL_W; ++q) Lc[q] = (q < w) ? P[q + (int64_t)c * h] : 0.0;
    const double dinv = 1.0 / P[c + (int64_t)c * h];
    double xi = (lane < w) ? x[f + lane] : 0.0;
    // y_j -= sum_i L21[i][j] x[rows[i]]: lanes along the (at most three chunks of) rows, one wave reduction per column
    const double xr0 = (lane < hu) ? x[rows[lane]] : 0.0;
    const double xr1 = (lane + 64 < hu) ? x[rows[lane + 64]] : 0.0;
    const double xr2 = (lane + 128 < hu) ? x[rows[lane + 128]] : 0.0;
    const double* __restrict__ P21 = P + w;
    const bool in0 = lane < hu, in1 = lane + 64 < hu, in2 = lane + 128 < hu;
    int j = 0;
    for (; j + 4 <= w; j += 4) {                         // four columns at a time: twelve loads, four reductions in flight
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double* __restrict__ col = P21 + (int64_t)(j + q) * h;
            const double c0 = in0 ? col[lane] : 0.0, c1 = in1 ? col[lane + 64] : 0.0, c2 = in2 ? col[lane + 128] : 0.0;
            v[q] = fma(c2, xr2, fma(c1, xr1, c0 * xr0));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], o, 64);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (lane == j + q) xi -= v[q];
    }
    for (; j < w; ++j) {
        const double* __restrict__ col = P21 + (int64_t)j * h;
        double v = in0 ? col[lane] * xr0 : 0.0;
        if (in1) v = fma(col[lane + 64], xr1, v);
        if (in2) v = fma(col[lane + 128], xr2, v);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == j) xi -= v;
    }
#pragma unroll
    for (int q = SP_SMALL_W - 1; q >= 0; --q) {
        if (q < w) {                                     // wave-uniform
            const double v = sp_bcast(xi * dinv, q);
            if (lane == q) xi = v;
            if (lane < q) xi = fma(-Lc[q], v, xi);
        }
    }
    if (lane < w) x[f + lane] = xi;
}

__global__ __launch_bounds__(256) void sp_bwd_small_kernel(SpDev d, int level_begin, int count, const double* __restrict__ panels,
                                                           double* __restrict__ x) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 4 + wave;
    if (k >= count) return;
    const int s = d.level_sn[level_begin + k];
    const int f = d.sn_first[s];
    const int w = d.sn_first[s + 1] - f;
    const int h = (int)(d.sn_rowptr[s + 1] - d.sn_rowptr[s]);
    sp_bwd_small_body(d, s, f, w, h, h - w, lane, panels, x);
}

// forward substitution, one workgroup per supernode of a level:  y_s = L11^-1 (b_s + children updates),
// then the front's right-hand-side remainder r_s = (children updates below) - L21 y_s is left for the parent
// (by sp_fwd_rem_kernel, many workgroups, when the panel is large).
__global__ __launch_bounds__(256) void sp_fwd_kernel(SpDev d, int level_begin, const double* __restrict__ panels,
                                                     double* __restrict__ x, double* __restrict__ rem,
                                                     const int64_t* __restrict__ rem_off, int64_t xstride,
                                                     int64_t remstride) {
    __shared__ double xs[256];
    x += (int64_t)blockIdx.y * xstride;          // several right-hand sides: one grid row each
    rem += (int64_t)blockIdx.y * remstride;
    const int s = d.level_sn[level_begin + blockIdx.x];
    const int tid = threadIdx.x;
    const int f = d.sn_first[s];
    const int w = d.sn_first[s + 1] - f;
    const int h = (int)(d.sn_rowptr[s + 1] - d.sn_rowptr[s]);
    const int hu = h - w;
    if (w > d.wide) return;                             // sp_fwd_wide_gather_kernel + launch_trsv_persistent + sp_fwd_rem_kernel
    if (sp_is_small(w, hu)) {                           // (a mixed level: the one-wave path on wave 0, the other waves leave)
        if (tid < 64) sp_fwd_small_body(d, s, f, w, h, hu, tid, panels, x, rem, rem_off, xs);
        return;
    }
    const double* __restrict__ P = panels + d.panel_off[s];
    double* __restrict__ R = rem + rem_off[s];          // hu entries
    for (int i = tid; i < hu; i += 256) R[i] = 0.0;
    if (tid < w) xs[tid] = x[f + tid];
    __syncthreads();
    for (int ci = d.child_ptr[s]; ci < d.child_ptr[s + 1]; ++ci) {
        const int c = d.child_list[ci];
        const int hc = (int)(d.sn_rowptr[c + 1] - d.sn_rowptr[c]) - (d.sn_first[c + 1] - d.sn_first[c]);
        const double* __restrict__ Rc = rem + rem_off[c];
        const int* __restrict__ rm = d.relmap + d.relmap_off[c];
        int i = tid;
        for (; i + 768 < hc; i += 1024) {                 // four independent (map, value) pairs in flight per thread
            int pq[4];
            double vq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { pq[q] = rm[i + 256 * q]; vq[q] = Rc[i + 256 * q]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int p = pq[q];
                if (p < w) xs[p] += vq[q];
                else R[p - w] += vq[q];
            }
        }
        for (; i < hc; i += 256) {
            const int p = rm[i];
            if (p < w) xs[p] += Rc[i];
            else R[p - w] += Rc[i];
        }
        __syncthreads();   // children one after the other: deterministic, no atomics
    }
    sp_trsv_fwd_lds(P, h, w, xs);
    if (tid < w) x[f + tid] = xs[tid];
    if ((int64_t)hu * w > SP_HEAVY) return;             // remainder by sp_fwd_rem_kernel
    for (int i = tid; i < hu; i += 256) {
        const double* __restrict__ Pi = P + w + i;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int j = 0;
        for (; j + 8 <= w; j += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = Pi[(int64_t)(j + q) * h];
            a0 += v[0] * xs[j];
            a1 += v[1] * xs[j + 1];
            a2 += v[2] * xs[j + 2];
            a3 += v[3] * xs[j + 3];
            a0 += v[4] * xs[j + 4];
            a1 += v[5] * xs[j + 5];
            a2 += v[6] * xs[j + 6];
            a3 += v[7] * xs[j + 7];
        }
        for (; j < w; ++j) a0 += Pi[(int64_t)j * h] * xs[j];
        R[i] -= (a0 + a1) + (a2 + a3);
    }
}

// R_s -= L21 y_s for the heavy supernodes of a level: grid (row chunks of 64, heavy supernodes[, right-hand sides]).  A
// workgroup owns 64 rows (one per lane); its four waves split the columns (j mod 4) and their partial sums are added in wave
// order, so the result does not depend on the schedule.  Any width: the solved block goes through LDS 256 entries at a time.
__global__ __launch_bounds__(256) void sp_fwd_rem_kernel(SpDev d, const int* __restrict__ heavy,
                                                         const double* __restrict__ panels, const double* __restrict__ x,
                                                         double* __restrict__ rem, const int64_t* __restrict__ rem_off,
                                                         int64_t xstride, int64_t remstride) {
    __shared__ double xs[256];
    __shared__ double red[4][64];
    x += (int64_t)blockIdx.z * xstride;
    rem += (int64_t)blockIdx.z * remstride;
    const int s = heavy[blockIdx.y];
    const int f = d.sn_first[s];
    const int w = d.sn_first[s + 1] - f;
    const int h = (int)(d.sn_rowptr[s + 1] - d.sn_rowptr[s]);
    const int hu = h - w;
    if ((int)blockIdx.x * 64 >= hu) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const double* __restrict__ P = panels + d.panel_off[s] + w + min(i, hu - 1);
    double a0 = 0.0, a1 = 0.0;
    for (int c0 = 0; c0 < w; c0 += 256) {
        const int wc = min(256, w - c0);
        __syncthreads();
        if ((int)threadIdx.x < wc) xs[threadIdx.x] = x[f + c0 + threadIdx.x];
        __syncthreads();
        const double* __restrict__ Pc = P + (int64_t)c0 * h;
        int j = wave;
        for (; j + 28 < wc; j += 32) {                  // eight loads in flight per thread (two were: the kernel is latency-bound)
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = Pc[(int64_t)(j + 4 * q) * h];
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                a0 += v[q] * xs[j + 4 * q];
                a1 += v[q + 1] * xs[j + 4 * q + 4];
            }
        }
        for (; j < wc; j += 4) a0 += Pc[(int64_t)j * h] * xs[j];
    }
    red[wave][lane] = a0 + a1;
    __syncthreads();
    if (wave == 0 && i < hu) rem[rem_off[s] + i] -= (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// children contributions of a wide supernode (w > SP_WIDE): x_s += updates into its columns, R_s = updates below them.
// Round 4: grid (64-position blocks of the front, wide supernodes of the level, right-hand sides), one wave per block.  The
// share of a child that lands in a block is a contiguous range of its rows (the extend-add boundary tables of the big fronts:
// every wide supernode is one); the wave spreads it over the block's 64 positions through LDS and every lane keeps the sum of
// ITS position in a register, children one after the other -- deterministic, no atomics, one global write per position.
// (One workgroup per supernode walking whole children was 15-50 us per level: two dependent loads per entry, 256 threads.)
__global__ __launch_bounds__(64) void sp_fwd_wide_gather_kernel(SpDev d, const int* __restrict__ wide, double* __restrict__ x,
                                                                double* __restrict__ rem, const int64_t* __restrict__ rem_off,
                                                                int64_t xstride, int64_t remstride) {
    __shared__ double val[64];
    x += (int64_t)blockIdx.z * xstride;
    rem += (int64_t)blockIdx.z * remstride;
    const int s = wide[blockIdx.y];
    const int lane = threadIdx.x;
    const int f = d.sn_first[s];
    const int w = d.sn_first[s + 1] - f;
    const int h = (int)(d.sn_rowptr[s + 1] - d.sn_rowptr[s]);
    const int p0 = blockIdx.x * 64;
    if (p0 >= h) return;
    const int p = p0 + lane;
    const int nbt = (h + 63) / 64;
    double acc = 0.0;
    for (int ci = d.child_ptr[s]; ci < d.child_ptr[s + 1]; ++ci) {
        const int c = d.child_list[ci];
        const int hc = (int)(d.sn_rowptr[c + 1] - d.sn_rowptr[c]) - (d.sn_first[c + 1] - d.sn_first[c]);
        if (hc <= 0) continue;
        const int* __restrict__ lb = d.ea_lb + d.ea_off[c];
        const int ia = lb[blockIdx.x], ib = lb[min((int)blockIdx.x + 1, nbt)];
        if (ib <= ia) continue;                              // (wave-uniform)
        const double* __restrict__ Rc = rem + rem_off[c];
        const int* __restrict__ rm = d.relmap + d.relmap_off[c];
        val[lane] = 0.0;
        __syncthreads();
        if (ia + lane < ib) val[rm[ia + lane] - p0] = Rc[ia + lane];    // at most 64 rows land in 64 positions
        __syncthreads();
        acc += val[lane];
        __syncthreads();
    }
    if (p < w) x[f + p] += acc;
    else if (p < h) rem[rem_off[s] + p - w] = acc;
}

// y_j -= sum_{i >= w} L[i][j] x[rows[i]] for `ncol` columns starting at j0: one wave per column, lanes along the
// (contiguous) column
__device__ __forceinline__ void sp_bwd_cols(const double* __restrict__ P, const int* __restrict__ rows, int h, int w,
                                            const double* __restrict__ x, int j0, int j1, double* out, int out_off) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // round 4: a wave takes FOUR consecutive columns at a time -- x[rows[i]] is gathered once for the four (the gather is the long
    // dependent load), eight column loads are in flight per lane -- instead of one column with four chunks of it
    for (int j = j0 + 4 * wave; j < j1; j += 16) {
        const int nc = min(4, j1 - j);
        const double* __restrict__ col = P + (int64_t)j * h;
        double acc[4] = {0.0, 0.0, 0.0, 0.0}, bcc[4] = {0.0, 0.0, 0.0, 0.0};
        int i = w + lane;
        if (nc == 4) {
            for (; i + 64 < h; i += 128) {
                const double xa = x[rows[i]], xb = x[rows[i + 64]];
                double va[4], vb[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { va[c] = col[i + (int64_t)c * h]; vb[c] = col[i + 64 + (int64_t)c * h]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) { acc[c] += va[c] * xa; bcc[c] += vb[c] * xb; }
            }
            for (; i < h; i += 64) {
                const double xa = x[rows[i]];
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] += col[i + (int64_t)c * h] * xa;
            }
        } else {
            for (; i < h; i += 64) {
                const double xa = x[rows[i]];
                for (int c = 0; c < nc; ++c) acc[c] += col[i + (int64_t)c * h] * xa;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += bcc[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
        }
        if (lane == 0)
            for (int c = 0; c < nc; ++c) out[j + c - out_off] -= acc[c];
    }
}

// heavy supernodes of a level, before sp_bwd_kernel: y_j -= sum_{i >= w} L[i][j] x[rows[i]].  Grid (groups of FOUR columns, heavy
// supernodes); the four waves of a workgroup split the rows (64-row chunks, wave = chunk mod 4; two chunks in flight per wave,
// x[rows[i]] gathered once for the four columns) and their partial sums are added in wave order: deterministic.  (Round 4: 16
// columns per workgroup with one wave per 4 columns walking ALL rows left the device at ~1.4 waves per SIMD, latency-bound.)
__global__ __launch_bounds__(256) void sp_bwd_gemv_kernel(SpDev d, const int* __restrict__ heavy,
                                                          const double* __restrict__ panels, double* __restrict__ x) {
    __shared__ double red[4][4];
    const int s = heavy[blockIdx.y];
    const int f = d.sn_first[s];
    const int w = d.sn_first[s + 1] - f;
    const int h = (int)(d.sn_rowptr[s + 1] - d.sn_rowptr[s]);
    const int j0 = blockIdx.x * 4;
    if (j0 >= w) return;
    const int nc = min(4, w - j0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* __restrict__ col = panels + d.panel_off[s] + (int64_t)j0 * h;
    const int* __restrict__ rows = d.sn_rows + d.sn_rowptr[s];
    double acc[4] = {0.0, 0.0, 0.0, 0.0}, bcc[4] = {0.0, 0.0, 0.0, 0.0};
    int i = w + 64 * wave + lane;
    if (nc == 4) {
        for (; i + 256 < h; i += 512) {
            const double xa = x[rows[i]], xb = x[rows[i + 256]];
            double va[4], vb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) { va[c] = col[i + (int64_t)c * h]; vb[c] = col[i + 256 + (int64_t)c * h]; }
#pragma unroll
            for (int c = 0; c < 4; ++c) { acc[c] += va[c] * xa; bcc[c] += vb[c] * xb; }
        }
        for (; i < h; i += 256) {
            const double xa = x[rows[i]];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += col[i + (int64_t)c * h] * xa;
        }
    } else {
        for (; i < h; i += 256) {
            const double xa = x[rows[i]];
            for (int c = 0; c < nc; ++c) acc[c] += col[i + (int64_t)c * h] * xa;
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] += bcc[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += __shfl_xor(acc[c], o, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) red[wave][c] = acc[c];
    }
    __syncthreads();
    if ((int)threadIdx.x < nc) {
        const int c = threadIdx.x;
        x[f + j0 + c] -= (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// backward substitution (root level first):  x_s = L11^-T (y_s - L21' x[rows below])
__global__ __launch_bounds__(256) void sp_bwd_kernel(SpDev d, int level_begin, const double* __restrict__ panels,
                                                     double* __restrict__ x) {
    __shared__ double xs[256];
    const int s = d.level_sn[level_begin + blockIdx.x];
    const int tid = threadIdx.x;
    const int f = d.sn_first[s];
    const int w = d.sn_first[s + 1] - f;
    const int h = (int)(d.sn_rowptr[s + 1] - d.sn_rowptr[s]);
    const double* __restrict__ P = panels + d.panel_off[s];
    if (w > d.wide) return;                             // sp_bwd_gemv_kernel + launch_trsv_persistent (transposed)
    if (sp_is_small(w, h - w)) {                        // (a mixed level: the one-wave path on wave 0)
        if (tid < 64) sp_bwd_small_body(d, s, f, w, h, h - w, tid, panels, x);
        return;
    }
    if (tid < w) xs[tid] = x[f + tid];
    __syncthreads();
    if ((int64_t)(h - w) * w <= SP_HEAVY && h > w) {
        sp_bwd_cols(P, d.sn_rows + d.sn_rowptr[s], h, w, x, 0, w, xs, 0);
        __syncthreads();
    }
    sp_trsv_bwd_lds(P, h, w, xs);
    if (tid < w) x[f + tid] = xs[tid];
}

__global__ void sp_permute_kernel(const double* __restrict__ in, double* __restrict__ out, const int* __restrict__ map, int n,
                                  int gather) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (gather) out[i] = in[map[i]];     // out[new] = in[perm[new]]
    else out[map[i]] = in[i];            // out[perm[new]] = in[new]
}

// y += G' (w .* z)  and  z_out = w .* (G x) - zs  for CSC G (HBM-bound, tiny)
__global__ __launch_bounds__(256) void sp_gemv_t_kernel(int n, const int64_t* __restrict__ cp, const int* __restrict__ ri,
                                                        const double* __restrict__ gv, const double* __restrict__ zss,
                                                        double* __restrict__ y) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int64_t k = cp[j]; k < cp[j + 1]; ++k) s += gv[k] * zss[ri[k]];
    y[j] += s;
}
__global__ __launch_bounds__(256) void sp_gemv_n_kernel(int m, const int64_t* __restrict__ rp, const int* __restrict__ ci,
                                                        const int* __restrict__ nzmap, const double* __restrict__ gv,
                                                        const double* __restrict__ x, const double* __restrict__ w,
                                                        const double* __restrict__ zs, double* __restrict__ z) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    double s = 0.0;
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) s += gv[nzmap[k]] * x[ci[k]];
    z[r] = w[r] * s - zs[r];
}
// y = A x for a CSR pattern whose values are looked up through nzmap (G by rows; H mirrored to a full symmetric CSR)
__global__ __launch_bounds__(256) void sp_spmv_kernel(int m, const int64_t* __restrict__ rp, const int* __restrict__ ci,
                                                      const int* __restrict__ nzmap, const double* __restrict__ vals,
                                                      const double* __restrict__ x, double* __restrict__ y) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    double s = 0.0;
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) s += vals[nzmap[k]] * x[ci[k]];
    y[r] = s;
}
__global__ void sp_scale2_kernel(const double* __restrict__ w, const double* __restrict__ z, double* __restrict__ zs,
                                 double* __restrict__ zss, int m) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        const double t = w[i] * z[i];
        zs[i] = t;
        zss[i] = w[i] * t;
    }
}

// ---- host-side driver object -----------------------------------------------------------------------------
template <class T>
static int up(T** d, const std::vector<T>& h) {
    KKT_HIP_CHECK(DEV_ALLOC(d, sizeof(T) * (h.size() ? h.size() : 1)));
    if (!h.empty()) KKT_HIP_CHECK(memcpy_sync(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    return 0;
}

int sparse_engine_create(SparseEngine& E, int n, int m, const int64_t* gcp, const int64_t* gri, const double* gv,
                         const int64_t* hcp, const int64_t* hri, const double* hv) {
    sparse_engine_free(E);
    if (int e = symbolic_analyze(E.sym, n, m, gcp, gri, hcp, hri)) return e;
    SparseSymbolic& S = E.sym;
    E.n = n;
    E.m = m;
    const int64_t gnnz = gcp[n], hnnz = hcp ? hcp[n] : 0;
    if (int e = up(&E.d_sn_first, S.sn_first)) return e;
    if (int e = up(&E.d_sn_rowptr, S.sn_rowptr)) return e;
    if (int e = up(&E.d_sn_rows, S.sn_rows)) return e;
    if (int e = up(&E.d_panel_off, S.panel_off)) return e;
    if (int e = up(&E.d_upd_off, S.upd_off)) return e;
    if (int e = up(&E.d_upd_ld, S.upd_ld)) return e;
    if (int e = up(&E.d_child_ptr, S.child_ptr)) return e;
    if (int e = up(&E.d_child_list, S.child_list)) return e;
    if (int e = up(&E.d_relmap_off, S.relmap_off)) return e;
    if (int e = up(&E.d_relmap, S.relmap)) return e;
    if (int e = up(&E.d_level_sn, S.level_sn)) return e;
    if (int e = up(&E.d_asm_slot, S.asm_slot)) return e;
    if (int e = up(&E.d_asm_ptr, S.asm_ptr)) return e;
    if (int e = up(&E.d_asm_a, S.asm_a)) return e;
    if (int e = up(&E.d_asm_b, S.asm_b)) return e;
    if (int e = up(&E.d_asm_r, S.asm_r)) return e;
    if (int e = up(&E.d_perm, S.perm)) return e;
    if (int e = up(&E.d_iperm, S.iperm)) return e;
    if (int e = up(&E.d_heavy, S.heavy)) return e;
    if (int e = up(&E.d_vb, S.vb)) return e;
    if (int e = up(&E.d_tv_tickets, S.tv_tickets)) return e;
    if (int e = up(&E.d_tv_prog_off, S.tv_prog_off)) return e;
    if (int e = up(&E.d_tv_linv_off, S.tv_linv_off)) return e;
    {   // state of the persistent tile kernel for every level, zeroed by ONE memset per factorisation
        size_t at = 0;
        E.tv_ctl_at.assign(S.nlevels, 0);
        E.tv_prog_at.assign(S.nlevels, 0);
        for (int l = 0; l < S.nlevels; ++l) {
            E.tv_ctl_at[l] = at;
            at += (potrf_tile_ctl_bytes() + 63) & ~(size_t)63;
            E.tv_prog_at[l] = at;
            at += (sizeof(unsigned) * 3 * (size_t)std::max(1, S.tv_nprog[l]) + 63) & ~(size_t)63;   // progress + half + micro words
        }
        E.tv_info_at = at;
        at += sizeof(int) * std::max<size_t>(1, S.vb.size());
        E.tv_state_bytes = at;
        KKT_HIP_CHECK(DEV_ALLOC(&E.d_tv_state, at));
    }
    KKT_HIP_CHECK(DEV_ALLOC(&E.d_tv_linv, sizeof(double) * 2048 * (size_t)std::max(1, S.tv_linv_max)));
    if (int e = potrf_work_init_batched(E.pw_vb, std::max(1, S.vb_maxcount))) return e;
    {   // G in CSC (values + int rows) and CSR (for G x)
        std::vector<double> gvals(gv, gv + gnnz), hvals;
        if (hv) hvals.assign(hv, hv + hnnz);
        std::vector<int64_t> cpv(gcp, gcp + n + 1);
        std::vector<int> riv(gnnz);
        for (int64_t k = 0; k < gnnz; ++k) riv[k] = (int)gri[k];
        std::vector<int64_t> rp(m + 1, 0);
        for (int64_t k = 0; k < gnnz; ++k) rp[gri[k] + 1]++;
        for (int r = 0; r < m; ++r) rp[r + 1] += rp[r];
        std::vector<int> ci(gnnz), nzmap(gnnz);
        std::vector<int64_t> pos(rp.begin(), rp.end() - 1);
        for (int j = 0; j < n; ++j)
            for (int64_t k = gcp[j]; k < gcp[j + 1]; ++k) {
                const int64_t p = pos[gri[k]]++;
                ci[p] = j;
                nzmap[p] = (int)k;
            }
        if (hcp) {   // H mirrored to a full symmetric CSR (values by reference to the lower-triangular CSC) for P x
            std::vector<int64_t> hrp(n + 1, 0);
            for (int j = 0; j < n; ++j)
                for (int64_t k = hcp[j]; k < hcp[j + 1]; ++k) {
                    const int i = (int)hri[k];
                    if (i < j) continue;
                    hrp[i + 1]++;
                    if (i != j) hrp[j + 1]++;
                }
            for (int r = 0; r < n; ++r) hrp[r + 1] += hrp[r];
            std::vector<int> hci(hrp[n]), hmap(hrp[n]);
            std::vector<int64_t> hpos(hrp.begin(), hrp.end() - 1);
            for (int j = 0; j < n; ++j)
                for (int64_t k = hcp[j]; k < hcp[j + 1]; ++k) {
                    const int i = (int)hri[k];
                    if (i < j) continue;
                    int64_t q = hpos[i]++;
                    hci[q] = j; hmap[q] = (int)k;
                    if (i != j) { q = hpos[j]++; hci[q] = i; hmap[q] = (int)k; }
                }
            if (int e = up(&E.d_hrp, hrp)) return e;
            if (int e = up(&E.d_hci, hci)) return e;
            if (int e = up(&E.d_hmap, hmap)) return e;
        }
        if (int e = up(&E.d_gv, gvals)) return e;
        if (int e = up(&E.d_hv, hvals)) return e;
        if (int e = up(&E.d_gcp, cpv)) return e;
        if (int e = up(&E.d_gri, riv)) return e;
        if (int e = up(&E.d_grp, rp)) return e;
        if (int e = up(&E.d_gci, ci)) return e;
        if (int e = up(&E.d_gnzmap, nzmap)) return e;
    }
    // remainders of the forward substitution: one (h - w) vector per supernode
    std::vector<int64_t> rem_off(S.ns + 1, 0);
    for (int s = 0; s < S.ns; ++s)
        rem_off[s + 1] = rem_off[s] + (S.sn_rowptr[s + 1] - S.sn_rowptr[s]) - (S.sn_first[s + 1] - S.sn_first[s]);
    if (int e = up(&E.d_rem_off, rem_off)) return e;
    KKT_HIP_CHECK(DEV_ALLOC(&E.d_rem, sizeof(double) * (rem_off[S.ns] ? rem_off[S.ns] : 1)));
    KKT_HIP_CHECK(DEV_ALLOC(&E.d_panels, sizeof(double) * (S.store_doubles ? S.store_doubles : 1)));
    {   // the panels as chunks of at most SP_ZERO_CHUNK doubles (sp_zero_chunks_kernel); the store as a whole starts defined: the
        // allocator cleared it (devmem.cpp) -- nothing but the panels and the lower triangles of the update matrices is ever read
        std::vector<int64_t> zoff;
        std::vector<int> zlen;
        for (int sn = 0; sn < S.ns; ++sn) {
            const int64_t hh = S.sn_rowptr[sn + 1] - S.sn_rowptr[sn], ww = S.sn_first[sn + 1] - S.sn_first[sn];
            for (int64_t a = 0; a < hh * ww; a += SP_ZERO_CHUNK) {
                zoff.push_back(S.panel_off[sn] + a);
                zlen.push_back((int)std::min<int64_t>(SP_ZERO_CHUNK, hh * ww - a));
            }
        }
        E.n_zero_chunks = (int)zoff.size();
        if (int e = up(&E.d_zero_off, zoff)) return e;
        if (int e = up(&E.d_zero_len, zlen)) return e;
    }
    E.d_upd = nullptr;          // update matrices live in the same buffer (offsets are absolute)
    KKT_HIP_CHECK(DEV_ALLOC(&E.d_xp, sizeof(double) * (n ? n : 1)));
    {   // the wide supernodes as jobs of the batched persistent triangular solve (single right-hand side: x = d_xp)
        std::vector<TrsvJob> jobs(S.wide.size());
        for (size_t k = 0; k < S.wide.size(); ++k) {
            const int s = S.wide[k];
            jobs[k] = TrsvJob{E.d_panels + S.panel_off[s], S.sn_rowptr[s + 1] - S.sn_rowptr[s], S.sn_first[s + 1] - S.sn_first[s], 0,
                              E.d_xp + S.sn_first[s]};
        }
        E.lvl_small.assign(S.nlevels, 0);
        for (int l = 0; l < S.nlevels; ++l)
            for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
                const int sn = S.level_sn[k];
                const int ww = S.sn_first[sn + 1] - S.sn_first[sn];
                const int hu = (int)(S.sn_rowptr[sn + 1] - S.sn_rowptr[sn]) - ww;
                if (ww <= SP_SMALL_W && hu <= SP_SMALL_HU) E.lvl_small[l]++;
            }
        E.wide_maxw.assign(S.nlevels, 0);
        for (int l = 0; l < S.nlevels; ++l)
            for (int k = S.wide_ptr[l]; k < S.wide_ptr[l + 1]; ++k)
                E.wide_maxw[l] = std::max(E.wide_maxw[l], S.sn_first[S.wide[k] + 1] - S.sn_first[S.wide[k]]);
        KKT_HIP_CHECK(DEV_ALLOC(&E.d_wide_jobs, sizeof(TrsvJob) * std::max<size_t>(1, jobs.size())));
        if (!jobs.empty()) KKT_HIP_CHECK(memcpy_sync(E.d_wide_jobs, jobs.data(), sizeof(TrsvJob) * jobs.size(), hipMemcpyHostToDevice));
        if (int e = up(&E.d_wide, S.wide)) return e;
        if (int e = up(&E.d_ea_off, S.ea_off)) return e;
        if (int e = up(&E.d_ea_lb, S.ea_lb)) return e;
    }
    KKT_HIP_CHECK(DEV_ALLOC(&E.d_info, sizeof(int)));
    KKT_HIP_CHECK(hipHostMalloc(&E.h_info, sizeof(int)));
    return 0;
}

void sparse_engine_free(SparseEngine& E) {
    void* ptrs[] = {E.d_sn_first, E.d_sn_rowptr, E.d_sn_rows, E.d_panel_off, E.d_upd_off, E.d_child_ptr, E.d_child_list,
                    E.d_relmap_off, E.d_relmap, E.d_level_sn, E.d_upd_ld, E.d_asm_slot, E.d_asm_ptr, E.d_asm_a, E.d_asm_b, E.d_asm_r,
                    E.d_perm, E.d_gv, E.d_hv, E.d_gcp, E.d_gri, E.d_grp, E.d_gci, E.d_gnzmap, E.d_rem_off, E.d_rem,
                    E.d_panels, E.d_upd, E.d_xp, E.d_info, E.d_heavy, E.d_vb, E.d_hrp, E.d_hci, E.d_hmap, E.d_rem_multi, E.d_iperm,
                    E.d_tv_tickets, E.d_tv_prog_off, E.d_tv_linv_off, E.d_tv_state, E.d_tv_linv, E.d_wide_jobs, E.d_wide, E.d_ea_off, E.d_ea_lb, E.d_zero_off, E.d_zero_len};
    for (void* p : ptrs)
        if (p) (void)dev_free(p);
    if (E.h_info) (void)hipHostFree(E.h_info);
    potrf_work_free(E.pw_vb);
    E = SparseEngine();
}

static SpDev devview(const SparseEngine& E) {
    SpDev d;
    d.sn_first = E.d_sn_first;
    d.sn_rowptr = E.d_sn_rowptr;
    d.sn_rows = E.d_sn_rows;
    d.panel_off = E.d_panel_off;
    d.upd_off = E.d_upd_off;
    d.upd_ld = E.d_upd_ld;
    d.child_ptr = E.d_child_ptr;
    d.child_list = E.d_child_list;
    d.relmap_off = E.d_relmap_off;
    d.relmap = E.d_relmap;
    d.level_sn = E.d_level_sn;
    d.ea_off = E.d_ea_off;
    d.ea_lb = E.d_ea_lb;
    d.wide = E.sym.wide_threshold;
    return d;
}

// numeric refactorisation with the current scaling di (device pointer); *info as LAPACK potrf (permuted column)
int sparse_engine_factor(SparseEngine& E, const double* d_di, hipStream_t st, int* info) {
    const SparseSymbolic& S = E.sym;
    if (E.n == 0) { if (info) *info = 0; return 0; }
    E.dense_root_level = -1;
    if (dev_knob("MI355KKT_SPARSE_POISON"))      // test knob: every byte of the store that is not cleared on purpose reads as NaN
        KKT_HIP_CHECK(hipMemsetAsync(E.d_panels, 0xff, sizeof(double) * (S.store_doubles ? S.store_doubles : 1), st));
    if (E.n_zero_chunks > 0)
        hipLaunchKernelGGL(sp_zero_chunks_kernel, dim3(E.n_zero_chunks), dim3(256), 0, st, E.d_zero_off, E.d_zero_len, E.d_panels);
    static const int imax = 0x7fffffff;      // (static: the asynchronous copy below must not read a dead stack slot on an early error return)
    KKT_HIP_CHECK(hipMemcpyAsync(E.d_info, &imax, sizeof(int), hipMemcpyHostToDevice, st));
    const int64_t nt = (int64_t)S.asm_slot.size();
    if (nt > 0)
        hipLaunchKernelGGL(sp_assemble_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, nt, E.d_asm_slot,
                           E.d_asm_ptr, E.d_asm_a, E.d_asm_b, E.d_asm_r, E.d_gv, E.d_hv, d_di, E.d_panels);
    const SpDev d = devview(E);
    const bool old_chain = dev_knob("MI355KKT_SPARSE_TILES") && !strcmp(dev_knob("MI355KKT_SPARSE_TILES"), "0");
    if (!old_chain) KKT_HIP_CHECK(hipMemsetAsync(E.d_tv_state, 0, E.tv_state_bytes, st));
    for (int l = 0; l < S.nlevels; ++l) {
        const int nsmall = S.level_nsmall[l];
        if (nsmall > 0)
            hipLaunchKernelGGL(sp_front_kernel, dim3(nsmall), dim3(256), 0, st, d, S.level_ptr[l], E.d_panels, E.d_panels,
                               E.d_info);
        const int nbig = S.vb_ptr[l + 1] - S.vb_ptr[l];
        if (nbig > 0) {
            // the big fronts of a level go through the dense MFMA kernels together (blockIdx.z = front)
            const VbDesc* dv = E.d_vb + S.vb_ptr[l];
            hipLaunchKernelGGL(sp_extend_add_vb_kernel, dim3((S.vb_maxh[l] + EA_COLS - 1) / EA_COLS, (S.vb_maxh[l] + EA_ROWS - 1) / EA_ROWS, nbig),
                               dim3(256), 0, st, d, dv, E.d_panels);
            // Round 4: a level whose only front is ONE big supernode without rows below it -- the root separator -- is a plain dense
            // Cholesky: the dense engine's persistent tile kernel runs it (no ragged-extent / ticket-table bookkeeping: 0 spilled
            // registers against the fronts' kernel's 61; 1.44 instead of 2.2 ms at 4096 columns) and leaves the 128 x 128 inverses
            // of its diagonal blocks for the solves (sp_wide_forward / _backward).
            const VbDesc& only = S.vb[S.vb_ptr[l]];
            const bool dense_root = !old_chain && nbig == 1 && nsmall == 0 && only.h == only.w && only.w >= 1024 &&
                                    (only.w + 127) / 128 <= 252 && !dev_knob("MI355KKT_SPARSE_NO_DENSE_ROOT");
            if (dense_root) {
                if (int e = launch_potrf(E.d_panels + only.off, only.h, only.w, E.pw_vb, st)) return e;
                hipLaunchKernelGGL(sp_merge_info_kernel, dim3(1), dim3(1), 0, st, E.pw_vb.d_info, only.col0, E.d_info);
                E.dense_root_level = l;
                // round 6: the root's solves in 512-row hops over all compute units (trsv512.hip) -- the 512 x 512 inverses of its
                // diagonal blocks from the 128 x 128 ones the tile kernel has just left
                const char* kw = dev_knob("MI355KKT_TRSV_WIDE");
                if (!(kw && atoi(kw) == 0) && E.t_num_cus > 0 && trsv_wide_rows(only.w, E.t_num_cus, true) != 0 &&
                    E.pw_vb.minv_n == only.w && E.pw_vb.minv_of == E.d_panels + only.off)
                    if (int e = launch_block_inverse512(E.d_panels + only.off, only.h, only.w, E.pw_vb, st)) return e;
            } else if (old_chain) {
                if (int e = launch_potrf_partial_vb(E.d_panels, dv, nbig, S.vb_maxh[l], S.vb_maxw[l], E.pw_vb, st)) return e;
                hipLaunchKernelGGL(sp_merge_info_vb_kernel, dim3((nbig + 255) / 256), dim3(256), 0, st, E.pw_vb.d_info, nbig, E.d_info);
            } else if (int e = launch_potrf_tiles_vb(E.d_panels, dv, nbig, E.d_tv_tickets + 4 * (size_t)S.tv_ptr[l],
                                                    S.tv_ptr[l + 1] - S.tv_ptr[l], E.d_tv_prog_off + S.vb_ptr[l],
                                                    E.d_tv_linv_off + S.vb_ptr[l], E.d_tv_state + E.tv_ctl_at[l],
                                                    reinterpret_cast<unsigned*>(E.d_tv_state + E.tv_prog_at[l]), S.tv_nprog[l], E.d_tv_linv,
                                                    reinterpret_cast<int*>(E.d_tv_state + E.tv_info_at) + S.vb_ptr[l], st, true))
                return e;
        }
    }
    if (!old_chain && !S.vb.empty())       // first failing pivot over the big fronts of all levels
        hipLaunchKernelGGL(sp_merge_info_vb_kernel, dim3(((unsigned)S.vb.size() + 255) / 256), dim3(256), 0, st,
                           reinterpret_cast<const int*>(E.d_tv_state + E.tv_info_at), (int)S.vb.size(), E.d_info);
    // the transposed persistent solve of the wide supernodes streams L11' from the (unused) upper triangle of the front
    // (one launch per level, its grid sized for THAT level's widest supernode: one launch for all of them sized for the root was
    //  4.1 M workgroups at 64^3, nearly all of them empty -- 0.4 ms of dispatch per factorisation)
    for (int l = 0; l < S.nlevels; ++l) {
        const int k0 = S.wide_ptr[l], nw = S.wide_ptr[l + 1] - k0;
        if (nw > 0)
            if (int e = launch_mirror_lower_jobs(E.d_wide_jobs + k0, nw, E.wide_maxw[l], st)) return e;
    }
    KKT_HIP_CHECK(hipGetLastError());
    KKT_HIP_CHECK(hipMemcpyAsync(E.h_info, E.d_info, sizeof(int), hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipStreamSynchronize(st));
    if (info) *info = (*E.h_info == imax) ? 0 : *E.h_info;
    return 0;
}

// The root supernode after a dense-root factorisation (sparse_engine_factor): its diagonal block was factored by the dense tile
// kernel, which left the 128 x 128 inverses in E.pw_vb -- the two-sweep solve of the dense engine when the order allows it,
// the one-sweep kernel with the inverses otherwise.
static int sp_root_solve(SparseEngine& E, int s, double* x, int trans, hipStream_t st) {
    const SparseSymbolic& S = E.sym;
    const int f = S.sn_first[s], w = S.sn_first[s + 1] - f;
    const int h = (int)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
    const double* P = E.d_panels + S.panel_off[s];
    const double* minv = (E.pw_vb.minv_n == w && E.pw_vb.minv_of == P) ? E.pw_vb.d_minv : nullptr;
    if (E.pw_vb.m512_n == w && E.pw_vb.m512_of == P && h == w)             // (formed by sparse_engine_factor when the order allows it)
        if (const int rows = trsv_wide_rows(w, E.t_num_cus, true))
            return launch_trsv_wide(P, h, w, x + f, trans, ++*E.t_epoch, E.t_err, st, E.pw_vb, rows, E.t_num_cus);
    if (minv && w % 128 == 0 && 2 * (w / 128) <= E.t_num_cus)
        return launch_trsv_pair(P, h, w, x + f, trans, ++*E.t_epoch, E.t_err, st, E.t_gran, minv);
    return launch_trsv_persistent(P, h, w, x + f, trans, ++*E.t_epoch, E.t_err, st, E.t_gran, minv);
}

// Wide supernodes of level l (w > SP_WIDE; the top separators): children contributions gathered in global memory, the dense
// diagonal block through the persistent multi-workgroup triangular solve of the dense engine (one right-hand side) or the
// blocked trsm (several).  Runs after the level's sp_fwd_kernel and before its sp_fwd_rem_kernel.
static int sp_wide_forward(SparseEngine& E, const SpDev& d, int l, double* x, double* rem, int64_t xstride, int64_t remstride,
                           int nrhs, hipStream_t st) {
    const SparseSymbolic& S = E.sym;
    const int k0 = S.wide_ptr[l], nw = S.wide_ptr[l + 1] - k0;
    if (nw == 0) return 0;
    int maxh = 0;
    for (int k = k0; k < k0 + nw; ++k) maxh = std::max(maxh, (int)(S.sn_rowptr[S.wide[k] + 1] - S.sn_rowptr[S.wide[k]]));
    hipLaunchKernelGGL(sp_fwd_wide_gather_kernel, dim3((maxh + 63) / 64, nw, nrhs), dim3(64), 0, st, d, E.d_wide + k0, x, rem,
                       E.d_rem_off, xstride, remstride);
    if (nrhs == 1 && E.t_gran && nw == 1 && l == E.dense_root_level)       // the dense root: inverses from the tile Cholesky
        return sp_root_solve(E, S.wide[k0], x, 0, st);
    if (nrhs == 1 && x == E.d_xp && E.t_gran && E.t_njobs_max > 0) {      // the level's systems, up to t_njobs_max per launch
        for (int c0 = 0; c0 < nw; c0 += E.t_njobs_max)
            if (int e = launch_trsv_persistent(nullptr, 0, E.wide_maxw[l], nullptr, 0, ++*E.t_epoch, E.t_err, st, E.t_gran,
                                               nullptr, E.d_wide_jobs + k0 + c0, std::min(E.t_njobs_max, nw - c0)))
                return e;
        return 0;
    }
    for (int k = k0; k < k0 + nw; ++k) {
        const int s = S.wide[k], f = S.sn_first[s], w = S.sn_first[s + 1] - f;
        const int h = (int)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
        const double* P = E.d_panels + S.panel_off[s];
        if (nrhs == 1 && E.t_gran) {
            if (int e = launch_trsv_persistent(P, h, w, x + f, 0, ++*E.t_epoch, E.t_err, st, E.t_gran)) return e;
        } else if (int e = launch_trsm_lower(P, h, w, x + f, xstride, nrhs, 0, st))
            return e;
    }
    return 0;
}
static int sp_wide_backward(SparseEngine& E, int l, double* x, hipStream_t st) {
    const SparseSymbolic& S = E.sym;
    const int k0 = S.wide_ptr[l], nw = S.wide_ptr[l + 1] - k0;
    if (nw == 0) return 0;
    if (E.t_gran && nw == 1 && l == E.dense_root_level) return sp_root_solve(E, S.wide[k0], x, 1, st);
    if (x == E.d_xp && E.t_gran && E.t_njobs_max > 0) {
        for (int c0 = 0; c0 < nw; c0 += E.t_njobs_max)
            if (int e = launch_trsv_persistent(nullptr, 0, E.wide_maxw[l], nullptr, 1, ++*E.t_epoch, E.t_err, st, E.t_gran,
                                               nullptr, E.d_wide_jobs + k0 + c0, std::min(E.t_njobs_max, nw - c0)))
                return e;
        return 0;
    }
    for (int k = S.wide_ptr[l]; k < S.wide_ptr[l + 1]; ++k) {
        const int s = S.wide[k], f = S.sn_first[s], w = S.sn_first[s + 1] - f;
        const int h = (int)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
        const double* P = E.d_panels + S.panel_off[s];
        if (E.t_gran) {      // (L' streamed from the mirrored upper triangle, see sparse_engine_factor)
            if (int e = launch_trsv_persistent(P, h, w, x + f, 1, ++*E.t_epoch, E.t_err, st, E.t_gran)) return e;
        } else if (int e = launch_trsm_lower(P, h, w, x + f, w, 1, 1, st))
            return e;
    }
    return 0;
}

// the level kernels of the supernodes that are not wide (E.lvl_small[l] = number of small supernodes of level l)
static void sp_launch_fwd_level(SparseEngine& E, const SpDev& d, int l, double* x, double* rem, int64_t xstride, int64_t remstride,
                                int nj, hipStream_t st) {
    const SparseSymbolic& S = E.sym;
    const int cnt = S.level_ptr[l + 1] - S.level_ptr[l];
    if (cnt <= 0) return;
    if (E.lvl_small[l] == cnt)                   // small supernodes only: four per workgroup
        hipLaunchKernelGGL(sp_fwd_small_kernel, dim3((cnt + 3) / 4, nj), dim3(256), 0, st, d, S.level_ptr[l], cnt, E.d_panels, x, rem,
                           E.d_rem_off, xstride, remstride);
    else                                         // (its small supernodes run the same one-wave body on one wave of their workgroup)
        hipLaunchKernelGGL(sp_fwd_kernel, dim3(cnt, nj), dim3(256), 0, st, d, S.level_ptr[l], E.d_panels, x, rem, E.d_rem_off, xstride,
                           remstride);
}
static void sp_launch_bwd_level(SparseEngine& E, const SpDev& d, int l, double* x, hipStream_t st) {
    const SparseSymbolic& S = E.sym;
    const int cnt = S.level_ptr[l + 1] - S.level_ptr[l];
    if (cnt <= 0) return;
    if (E.lvl_small[l] == cnt)
        hipLaunchKernelGGL(sp_bwd_small_kernel, dim3((cnt + 3) / 4), dim3(256), 0, st, d, S.level_ptr[l], cnt, E.d_panels, x);
    else
        hipLaunchKernelGGL(sp_bwd_kernel, dim3(cnt), dim3(256), 0, st, d, S.level_ptr[l], E.d_panels, x);
}

// forward half: E.d_xp := L^-1 P b (b = d_in, original ordering); optionally copied to d_out_perm (permuted ordering)
int sparse_engine_forward(SparseEngine& E, const double* d_in, double* d_out_perm, hipStream_t st) {
    const SparseSymbolic& S = E.sym;
    if (E.n == 0) return 0;
    const SpDev d = devview(E);
    const dim3 g((E.n + 255) / 256);
    hipLaunchKernelGGL(sp_permute_kernel, g, dim3(256), 0, st, d_in, E.d_xp, E.d_perm, E.n, 1);
    for (int l = 0; l < S.nlevels; ++l) {
        sp_launch_fwd_level(E, d, l, E.d_xp, E.d_rem, 0, 0, 1, st);
        if (int e = sp_wide_forward(E, d, l, E.d_xp, E.d_rem, 0, 0, 1, st)) return e;
        const int nh = S.heavy_ptr[l + 1] - S.heavy_ptr[l];
        if (nh > 0)
            hipLaunchKernelGGL(sp_fwd_rem_kernel, dim3((S.heavy_maxhu[l] + 63) / 64, nh), dim3(256), 0, st, d,
                               E.d_heavy + S.heavy_ptr[l], E.d_panels, E.d_xp, E.d_rem, E.d_rem_off, (int64_t)0, (int64_t)0);
    }
    if (d_out_perm) KKT_HIP_CHECK(hipMemcpyAsync(d_out_perm, E.d_xp, sizeof(double) * E.n, hipMemcpyDeviceToDevice, st));
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}
// out (n x nrhs, permuted ordering, ld n) := L^-1 P A' for the nrhs rows of the dense A (nrhs x n, ld lda): all right-hand
// sides go through the level-scheduled forward substitution together (one grid row per right-hand side)
__global__ __launch_bounds__(256) void sp_gather_rows_kernel(const double* __restrict__ A, int64_t lda, const int* __restrict__ perm,
                                                             int n, double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i < n) out[i + (int64_t)j * n] = A[j + (int64_t)perm[i] * lda];
}
int sparse_engine_forward_rows(SparseEngine& E, const double* d_A, int64_t lda, int nrhs, double* d_out, hipStream_t st) {
    const SparseSymbolic& S = E.sym;
    if (E.n == 0 || nrhs <= 0) return 0;
    const SpDev d = devview(E);
    const int64_t remtot = S.sn_rowptr[S.ns] - (int64_t)E.n;        // sum over supernodes of (h - w)
    if (E.rem_multi_cols < nrhs) {
        if (E.d_rem_multi) (void)dev_free(E.d_rem_multi);
        E.d_rem_multi = nullptr;
        KKT_HIP_CHECK(DEV_ALLOC(&E.d_rem_multi, sizeof(double) * (size_t)(remtot > 0 ? remtot : 1) * nrhs));
        E.rem_multi_cols = nrhs;
    }
    for (int j0 = 0; j0 < nrhs; j0 += 65535) {                      // grid.y / grid.z limit
        const int nj = std::min(65535, nrhs - j0);
        double* out = d_out + (size_t)j0 * E.n;
        hipLaunchKernelGGL(sp_gather_rows_kernel, dim3((E.n + 255) / 256, nj), dim3(256), 0, st, d_A + j0, lda, E.d_perm, E.n, out);
        for (int l = 0; l < S.nlevels; ++l) {
            sp_launch_fwd_level(E, d, l, out, E.d_rem_multi, (int64_t)E.n, remtot, nj, st);
            if (int e = sp_wide_forward(E, d, l, out, E.d_rem_multi, (int64_t)E.n, remtot, nj, st)) return e;
            const int nh = S.heavy_ptr[l + 1] - S.heavy_ptr[l];
            if (nh > 0)
                hipLaunchKernelGGL(sp_fwd_rem_kernel, dim3((S.heavy_maxhu[l] + 63) / 64, nh, nj), dim3(256), 0, st, d,
                                   E.d_heavy + S.heavy_ptr[l], E.d_panels, out, E.d_rem_multi, E.d_rem_off, (int64_t)E.n, remtot);
        }
    }
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}
// The same for a SPARSE A (CSR on the device: row r = right-hand side r): reference misc.py:1483-1487 keeps A' sparse through
// cholmod.spsolve (src/C/cholmod.c:583-654); here every row is scattered straight into its permuted dense right-hand side
// (no dense copy of A anywhere) and the rows go through the level-scheduled forward substitution `chunk` at a time, so the
// transient workspace stays bounded for any number of equality constraints.  d_out: n x nrhs, ld n.
__global__ __launch_bounds__(256) void sp_scatter_rows_csr_kernel(const int64_t* __restrict__ rp, const int* __restrict__ ci,
                                                                  const double* __restrict__ v, int r0, const int* __restrict__ iperm,
                                                                  int n, double* __restrict__ out) {
    const int j = blockIdx.x;                           // right-hand side j <- row r0 + j of A
    double* o = out + (int64_t)j * n;
    for (int i = threadIdx.x; i < n; i += 256) o[i] = 0.0;
    __syncthreads();
    const int64_t a = rp[r0 + j], b = rp[r0 + j + 1];
    for (int64_t k = a + threadIdx.x; k < b; k += 256) o[iperm[ci[k]]] = v[k];
}
int sparse_engine_forward_rows_csr(SparseEngine& E, const int64_t* d_rp, const int* d_ci, const double* d_v, int nrhs, double* d_out,
                                   hipStream_t st, int chunk) {
    const SparseSymbolic& S = E.sym;
    if (E.n == 0 || nrhs <= 0) return 0;
    if (chunk <= 0) chunk = 256;
    if (chunk > 65535) chunk = 65535;
    const SpDev d = devview(E);
    const int64_t remtot = S.sn_rowptr[S.ns] - (int64_t)E.n;
    const int need = std::min(chunk, nrhs);
    if (E.rem_multi_cols < need) {
        if (E.d_rem_multi) (void)dev_free(E.d_rem_multi);
        E.d_rem_multi = nullptr;
        E.rem_multi_cols = 0;
        KKT_HIP_CHECK(DEV_ALLOC(&E.d_rem_multi, sizeof(double) * (size_t)(remtot > 0 ? remtot : 1) * need));
        E.rem_multi_cols = need;
    }
    for (int j0 = 0; j0 < nrhs; j0 += chunk) {
        const int nj = std::min(chunk, nrhs - j0);
        double* out = d_out + (size_t)j0 * E.n;
        hipLaunchKernelGGL(sp_scatter_rows_csr_kernel, dim3(nj), dim3(256), 0, st, d_rp, d_ci, d_v, j0, E.d_iperm, E.n, out);
        for (int l = 0; l < S.nlevels; ++l) {
            sp_launch_fwd_level(E, d, l, out, E.d_rem_multi, (int64_t)E.n, remtot, nj, st);
            if (int e = sp_wide_forward(E, d, l, out, E.d_rem_multi, (int64_t)E.n, remtot, nj, st)) return e;
            const int nh = S.heavy_ptr[l + 1] - S.heavy_ptr[l];
            if (nh > 0)
                hipLaunchKernelGGL(sp_fwd_rem_kernel, dim3((S.heavy_maxhu[l] + 63) / 64, nh, nj), dim3(256), 0, st, d,
                                   E.d_heavy + S.heavy_ptr[l], E.d_panels, out, E.d_rem_multi, E.d_rem_off, (int64_t)E.n, remtot);
        }
    }
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}
// backward half: d_out := P' L^-T E.d_xp
int sparse_engine_backward(SparseEngine& E, double* d_out, hipStream_t st) {
    const SparseSymbolic& S = E.sym;
    if (E.n == 0) return 0;
    const SpDev d = devview(E);
    const dim3 g((E.n + 255) / 256);
    for (int l = S.nlevels - 1; l >= 0; --l) {
        const int nh = S.heavy_ptr[l + 1] - S.heavy_ptr[l];
        if (nh > 0)
            hipLaunchKernelGGL(sp_bwd_gemv_kernel, dim3((S.heavy_maxw[l] + 3) / 4, nh), dim3(256), 0, st, d,
                               E.d_heavy + S.heavy_ptr[l], E.d_panels, E.d_xp);
        sp_launch_bwd_level(E, d, l, E.d_xp, st);
        if (int e = sp_wide_backward(E, l, E.d_xp, st)) return e;
    }
    hipLaunchKernelGGL(sp_permute_kernel, g, dim3(256), 0, st, E.d_xp, d_out, E.d_perm, E.n, 0);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}
// x := S^-1 x  (x: device vector of length n, original ordering)
int sparse_engine_solve(SparseEngine& E, double* d_x, hipStream_t st) {
    if (int e = sparse_engine_forward(E, d_x, nullptr, st)) return e;
    return sparse_engine_backward(E, d_x, st);
}

// the two sparse products of solve():  zs = w.*z, x += G'(w.*zs)   and   z = w.*(G x) - zs
int sparse_engine_gemv_t(SparseEngine& E, const double* d_w, const double* d_z, double* d_zs, double* d_zss, double* d_x,
                         hipStream_t st) {
    if (E.m > 0) hipLaunchKernelGGL(sp_scale2_kernel, dim3((E.m + 255) / 256), dim3(256), 0, st, d_w, d_z, d_zs, d_zss, E.m);
    if (E.n > 0 && E.m > 0)
        hipLaunchKernelGGL(sp_gemv_t_kernel, dim3((E.n + 255) / 256), dim3(256), 0, st, E.n, E.d_gcp, E.d_gri, E.d_gv, d_zss,
                           d_x);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}
int sparse_engine_gemv_n(SparseEngine& E, const double* d_w, const double* d_x, const double* d_zs, double* d_z,
                         hipStream_t st) {
    if (E.m > 0)
        hipLaunchKernelGGL(sp_gemv_n_kernel, dim3((E.m + 255) / 256), dim3(256), 0, st, E.m, E.d_grp, E.d_gci, E.d_gnzmap,
                           E.d_gv, d_x, d_w, d_zs, d_z);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

// one product with the CSC / CSR copies: which = 0: G (trans: G'), 2: H (symmetric)
int sparse_engine_product(SparseEngine& E, int which, int trans, const double* d_in, double* d_out, hipStream_t st) {
    const dim3 gn((E.n + 255) / 256), gm((E.m + 255) / 256);
    if (which == 0 && !trans) {
        if (E.m > 0) hipLaunchKernelGGL(sp_spmv_kernel, gm, dim3(256), 0, st, E.m, E.d_grp, E.d_gci, E.d_gnzmap, E.d_gv, d_in, d_out);
    } else if (which == 0) {
        KKT_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(double) * E.n, st));
        if (E.m > 0) hipLaunchKernelGGL(sp_gemv_t_kernel, gn, dim3(256), 0, st, E.n, E.d_gcp, E.d_gri, E.d_gv, d_in, d_out);
    } else if (E.d_hrp) {
        hipLaunchKernelGGL(sp_spmv_kernel, gn, dim3(256), 0, st, E.n, E.d_hrp, E.d_hci, E.d_hmap, E.d_hv, d_in, d_out);
    } else {
        KKT_HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(double) * E.n, st));
    }
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

// residual products of the interior-point loop (coneprog.py:2170-2186): Gx = G x, GTz = G' z, Px = H x
int sparse_engine_products(SparseEngine& E, const double* d_x, const double* d_z, double* d_Gx, double* d_GTz, double* d_Px,
                           hipStream_t st) {
    const dim3 gn((E.n + 255) / 256), gm((E.m + 255) / 256);
    if (E.m > 0)
        hipLaunchKernelGGL(sp_spmv_kernel, gm, dim3(256), 0, st, E.m, E.d_grp, E.d_gci, E.d_gnzmap, E.d_gv, d_x, d_Gx);
    KKT_HIP_CHECK(hipMemsetAsync(d_GTz, 0, sizeof(double) * E.n, st));
    if (E.m > 0) hipLaunchKernelGGL(sp_gemv_t_kernel, gn, dim3(256), 0, st, E.n, E.d_gcp, E.d_gri, E.d_gv, d_z, d_GTz);
    if (E.d_hrp)
        hipLaunchKernelGGL(sp_spmv_kernel, gn, dim3(256), 0, st, E.n, E.d_hrp, E.d_hci, E.d_hmap, E.d_hv, d_x, d_Px);
    else
        KKT_HIP_CHECK(hipMemsetAsync(d_Px, 0, sizeof(double) * E.n, st));
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mi355kkt
