// Fill-reducing orderings for the sparse engine: nested dissection with minimum-degree leaves, approximate minimum
// degree, elimination-tree postorder, and the cost model (nnz(L), flops) that picks between them.  Host only.
//
//   * approximate minimum degree: quotient-graph elimination with element absorption, approximate external degrees,
//     mass elimination and supervariable (indistinguishable node) detection, after Amestoy, Davis & Duff, "An
//     approximate minimum degree ordering algorithm", SIAM J. Matrix Anal. Appl. 17 (1996) -- written from the paper.
//   * nested dissection: George's automatic scheme (BFS level structures from a pseudo-peripheral node); the chosen
//     level is thinned to a minimal vertex separator; parts below a size limit are ordered by the minimum-degree
//     routine with the already-placed separators as a halo (constrained minimum degree).
#include "ordering.h"
#include "knobs.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <chrono>
#include <functional>
#include <future>
#include <memory>
#include <numeric>
#include <system_error>

namespace mi355kkt {

// =====================================================================================================
// elimination tree, column counts, postorder
// =====================================================================================================
// Column counts by the skeleton / least-common-ancestor algorithm of Gilbert, Ng & Peyton ("An efficient algorithm to
// compute row and column counts for sparse Cholesky factorization", SIMAX 1994; the formulation of Davis, "Direct
// Methods for Sparse Linear Systems", section 4.5): O(nnz(A) alpha) instead of the O(nnz(L)) row-subtree walk -- the
// counts feed the cost model of two candidate orderings, so they are on the analysis' critical path.
void etree_and_counts(const Graph& adj, const std::vector<int>& order, std::vector<int>& parent, std::vector<int64_t>& cc) {
    const int n = (int)adj.size();
    std::vector<int> iperm(n);
    for (int k = 0; k < n; ++k) iperm[order[k]] = k;
    parent.assign(n, -1);
    cc.assign(n, 0);
    std::vector<int> anc(n, -1);
    // Liu's algorithm with path compression, row by row; rows in permuted order
    for (int i = 0; i < n; ++i)
        for (int u : adj[order[i]]) {
            int r = iperm[u];
            if (r >= i) continue;
            while (anc[r] != -1 && anc[r] != i) {
                const int nx = anc[r];
                anc[r] = i;
                r = nx;
            }
            if (anc[r] == -1) {
                anc[r] = i;
                parent[r] = i;
            }
        }
    // a postorder of the tree (children in index order)
    std::vector<int> post;
    post.reserve(n);
    {
        std::vector<int> head(n, -1), next(n, -1), stack;
        for (int v = n - 1; v >= 0; --v)
            if (parent[v] >= 0) { next[v] = head[parent[v]]; head[parent[v]] = v; }
        for (int r = 0; r < n; ++r) {
            if (parent[r] >= 0) continue;
            stack.push_back(r);
            while (!stack.empty()) {
                const int v = stack.back(), c = head[v];
                if (c >= 0) { head[v] = next[c]; stack.push_back(c); }
                else { post.push_back(v); stack.pop_back(); }
            }
        }
    }
    // first[j] = postorder rank of the first descendant of j; delta[j] = 1 for the leaves of the tree
    std::vector<int> first(n, -1), maxfirst(n, -1), prevleaf(n, -1), ancestor(n);
    std::vector<int64_t>& delta = cc;
    for (int k = 0; k < n; ++k) {
        int j = post[k];
        delta[j] = (first[j] == -1) ? 1 : 0;
        for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k;
    }
    for (int i = 0; i < n; ++i) ancestor[i] = i;
    for (int k = 0; k < n; ++k) {
        const int j = post[k];
        if (parent[j] != -1) delta[parent[j]]--;          // j is not a root
        for (int u : adj[order[j]]) {
            const int i = iperm[u];
            // is j a leaf of the i-th row subtree?
            if (i <= j || first[j] <= maxfirst[i]) continue;
            maxfirst[i] = first[j];
            const int jprev = prevleaf[i];
            prevleaf[i] = j;
            delta[j]++;                                    // j is a (first or subsequent) leaf: one more entry in column j
            if (jprev != -1) {                             // subsequent leaf: the overlap starts at lca(jprev, j)
                int q = jprev;
                while (q != ancestor[q]) q = ancestor[q];
                for (int sx = jprev; sx != q;) {
                    const int sp = ancestor[sx];
                    ancestor[sx] = q;
                    sx = sp;
                }
                delta[q]--;
            }
        }
        if (parent[j] != -1) ancestor[j] = parent[j];
    }
    for (int k = 0; k < n; ++k) {                          // sum the deltas up the tree, children before parents
        const int j = post[k];
        if (parent[j] != -1) cc[parent[j]] += cc[j];
    }
    for (int j = 0; j < n; ++j) cc[j] -= 1;                // entries below the diagonal
}

// the same counts by walking every row subtree (each entry of L is visited once): the check of the fast algorithm
void column_counts_by_row_subtrees(const Graph& adj, const std::vector<int>& order, const std::vector<int>& parent,
                                   std::vector<int64_t>& cc) {
    const int n = (int)adj.size();
    std::vector<int> iperm(n), mark(n, -1);
    for (int k = 0; k < n; ++k) iperm[order[k]] = k;
    cc.assign(n, 0);
    for (int i = 0; i < n; ++i) {
        mark[i] = i;
        for (int u : adj[order[i]]) {
            const int k = iperm[u];
            if (k >= i) continue;
            for (int j = k; j != -1 && j < i && mark[j] != i; j = parent[j]) {
                mark[j] = i;
                cc[j]++;
            }
        }
    }
}

namespace {

// postorder of the forest `parent` (children before parents, every subtree contiguous); among the children of a node
// the one with the largest column count comes last, so that it is adjacent to its parent (supernode chains)
void postorder(const std::vector<int>& parent, const std::vector<int64_t>& cc, std::vector<int>& post) {
    const int n = (int)parent.size();
    std::vector<int> head(n, -1), next(n, -1), roots;
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    // children are pushed at the front of their parent's list in decreasing count order, so every list runs from the
    // smallest to the largest count and is visited in that order
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cc[a] > cc[b]; });
    for (int v : idx) {
        if (parent[v] < 0) { roots.push_back(v); continue; }
        next[v] = head[parent[v]];
        head[parent[v]] = v;        // head = smallest cc ... tail = largest cc
    }
    std::sort(roots.begin(), roots.end());
    post.clear();
    post.reserve(n);
    std::vector<int> stack;
    for (int r : roots) {
        stack.push_back(r);
        while (!stack.empty()) {
            const int v = stack.back();
            const int c = head[v];
            if (c >= 0) {
                head[v] = next[c];   // consume child
                stack.push_back(c);
            } else {
                post.push_back(v);
                stack.pop_back();
            }
        }
    }
}

void cost_of(const std::vector<int64_t>& cc, int64_t& nnz, double& flops) {
    nnz = 0;
    flops = 0.0;
    for (int64_t c : cc) {
        nnz += c + 1;
        flops += (double)(c + 1) * (double)(c + 1);
    }
}

}  // namespace

// Supernode partition of a POSTORDERED elimination tree: column j joins the supernode that ends at j-1 when
// parent[j-1] == j (so struct(j-1) \ {j} is contained in struct(j)) and the explicit zeros this adds to the stored panel
// stay a small fraction of it.  Exact (fundamental) merges add none; relaxed merges trade a little fill for far fewer,
// denser fronts and a much shallower supernodal tree (other subtrees may hang off any column of the supernode).
void relaxed_supernodes(const std::vector<int>& parent, const std::vector<int64_t>& cc, std::vector<int>& sn_first,
                        std::vector<int>& sn_of) {
    const int n = (int)parent.size();
    // widest supernode: wider ones (the top separators of a nested dissection) are factored and solved by the multi-workgroup
    // dense kernels as ONE front instead of a chain of 256-column pieces, one level each ($MI355KKT_SN_MAXW: experiments)
    const int MAXW = dev_knob("MI355KKT_SN_MAXW") ? std::max(1, atoi(dev_knob("MI355KKT_SN_MAXW"))) : 8192;   // (read per analysis)
    sn_first.clear();
    sn_of.assign(n, 0);
    int64_t true_nnz = 0;        // nonzeros of L in the columns of the current supernode
    for (int j = 0; j < n; ++j) {
        bool join = false;
        const int64_t cj = 1 + cc[j];
        if (j > 0 && parent[j - 1] == j) {
            const int first = sn_first.back();
            const int64_t wn = j - first + 1;
            const int64_t stored = wn * cc[j] + wn * (wn + 1) / 2;   // trapezoid ending at column j
            const int64_t zeros = stored - (true_nnz + cj);
            const double lim = wn <= 4 ? 0.5 : (wn <= 16 ? 0.3 : (wn <= 64 ? 0.2 : 0.1));
            if (wn <= MAXW && (zeros == 0 || (double)zeros <= lim * (double)stored)) join = true;
        }
        if (!join) { sn_first.push_back(j); true_nnz = 0; }
        true_nnz += cj;
        sn_of[j] = (int)sn_first.size() - 1;
    }
}

namespace {

// height of the supernodal elimination tree of a postordered tree: the number of dependent steps of the
// level-scheduled device factorisation and solves
int supernodal_height(const std::vector<int>& parent, const std::vector<int64_t>& cc) {
    const int n = (int)parent.size();
    std::vector<int> sn_first, sn;
    relaxed_supernodes(parent, cc, sn_first, sn);
    const int ns = (int)sn_first.size();
    std::vector<int> h(ns, 1);
    int best = ns ? 1 : 0;
    for (int j = 0; j < n; ++j) {
        const int p = parent[j];
        if (p < 0 || sn[p] == sn[j]) continue;
        h[sn[p]] = std::max(h[sn[p]], h[sn[j]] + 1);
        best = std::max(best, h[sn[p]]);
    }
    return best;
}

}  // namespace

// =====================================================================================================
// approximate minimum degree
// =====================================================================================================
void amd_order(const Graph& adj, const std::vector<int>& nodes, const std::vector<int>& halo, std::vector<int>& local,
               int* out) {
    const int nf = (int)nodes.size(), N = nf + (int)halo.size();
    if (nf == 0) return;
    for (int k = 0; k < nf; ++k) local[nodes[k]] = k;
    for (size_t k = 0; k < halo.size(); ++k) local[halo[k]] = nf + (int)k;
    auto gid = [&](int i) { return i < nf ? nodes[i] : halo[i - nf]; };

    enum : unsigned char { VAR = 0, ELEM = 1, DEAD = 2 };
    std::vector<std::vector<int>> A(N), E(N), L(N);
    std::vector<int> nv(N, 1), deg(N, 0), member(N, -1), last_member(N);
    std::vector<unsigned char> state(N, VAR);
    std::vector<int64_t> w(N, 0), Lsize(N, 0);
    std::vector<int> stamp(N, -1), tag(N, -1);
    for (int i = 0; i < N; ++i) {
        last_member[i] = i;
        for (int u : adj[gid(i)]) {
            const int j = local[u];
            if (j < 0 || j == i) continue;
            if (i >= nf && j >= nf) continue;   // halo-halo edges never influence a free node's degree
            A[i].push_back(j);
        }
        deg[i] = (int)A[i].size();
    }
    // degree buckets over the free nodes
    std::vector<int> head(N + 1, -1), nxt(N, -1), prv(N, -1);
    auto bucket_insert = [&](int i) {
        const int d = deg[i];
        nxt[i] = head[d];
        prv[i] = -1;
        if (head[d] >= 0) prv[head[d]] = i;
        head[d] = i;
    };
    auto bucket_remove = [&](int i) {
        if (prv[i] >= 0) nxt[prv[i]] = nxt[i];
        else head[deg[i]] = nxt[i];
        if (nxt[i] >= 0) prv[nxt[i]] = prv[i];
        nxt[i] = prv[i] = -1;
    };
    for (int i = 0; i < nf; ++i) bucket_insert(i);

    int mindeg = 0, nout = 0;
    int64_t wflg = 1;
    int nleft = N;                       // weight of the variables not yet eliminated (halo included)
    std::vector<int> Lp, hashed;
    std::vector<int64_t> hsh(N, 0);
    int cur = 0, tcur = 0;
    auto emit_members = [&](int p) {     // p and everything merged into it, in merge order
        for (int v = p; v != -1; v = member[v]) out[nout++] = gid(v);
    };

    while (nout < nf) {
        while (mindeg <= N && head[mindeg] < 0) ++mindeg;
        if (mindeg > N) break;           // cannot happen: free weight left implies a listed node
        const int p = head[mindeg];
        bucket_remove(p);
        // ---- the new element: Lp = (A_p U union of L_e, e in E_p) \ {p}
        ++cur;
        Lp.clear();
        stamp[p] = cur;
        int64_t degLp = 0;
        for (int i : A[p])
            if (state[i] == VAR && stamp[i] != cur) { stamp[i] = cur; Lp.push_back(i); degLp += nv[i]; }
        for (int e : E[p]) {
            if (state[e] != ELEM) continue;
            for (int i : L[e])
                if (state[i] == VAR && stamp[i] != cur) { stamp[i] = cur; Lp.push_back(i); degLp += nv[i]; }
            state[e] = DEAD;             // absorbed into p
            std::vector<int>().swap(L[e]);
        }
        std::vector<int>().swap(A[p]);
        std::vector<int>().swap(E[p]);
        state[p] = ELEM;
        int nvp = nv[p];
        nleft -= nvp;
        // ---- |L_e \ Lp| for every element adjacent to a variable of Lp
        const int64_t wbase = wflg;
        int64_t wmax = 0;
        for (int i : Lp)
            for (int e : E[i]) {
                if (state[e] != ELEM) continue;
                if (w[e] < wbase) { w[e] = Lsize[e] + wbase; wmax = std::max(wmax, Lsize[e]); }
                w[e] -= nv[i];
            }
        wflg = wbase + wmax + 1;
        // ---- update the variables of Lp
        hashed.clear();
        for (int i : Lp) {
            if (i < nf) bucket_remove(i);
            // prune E_i and A_i in place
            int64_t d = 0, h = 0;
            std::vector<int>& Ei = E[i];
            std::vector<int>& Ai = A[i];
            size_t ne = 0, na = 0;
            for (int e : Ei) {
                if (state[e] != ELEM) continue;
                const int64_t dext = w[e] - wbase;
                if (dext > 0) { Ei[ne++] = e; d += dext; h += e; }
                else state[e] = DEAD, std::vector<int>().swap(L[e]);     // L_e is inside Lp: aggressive absorption
            }
            for (int j : Ai)
                if (state[j] == VAR && stamp[j] != cur) { Ai[na++] = j; d += nv[j]; h += j; }
            if (na == 0 && ne == 0 && i < nf) {
                // mass elimination: i is adjacent to nothing but the new element
                member[last_member[p]] = i;
                last_member[p] = last_member[i];
                nvp += nv[i];
                degLp -= nv[i];
                nleft -= nv[i];
                nv[i] = 0;
                state[i] = DEAD;
                std::vector<int>().swap(A[i]);
                std::vector<int>().swap(E[i]);
                continue;
            }
            Ei.resize(ne);
            Ei.push_back(p);
            Ai.resize(na);
            h += p;
            deg[i] = (int)std::min<int64_t>(deg[i], d);      // + |Lp \ i| below, once degLp is final
            hsh[i] = h % N;
            hashed.push_back(i);
        }
        // ---- supervariables: variables of Lp with identical adjacency in the quotient graph
        std::sort(hashed.begin(), hashed.end(), [&](int a, int b) { return hsh[a] != hsh[b] ? hsh[a] < hsh[b] : a < b; });
        for (size_t a = 0; a < hashed.size(); ++a) {
            const int i = hashed[a];
            if (state[i] != VAR) continue;
            bool tagged = false;
            for (size_t b = a + 1; b < hashed.size() && hsh[hashed[b]] == hsh[i]; ++b) {
                const int j = hashed[b];
                if (state[j] != VAR || (i < nf) != (j < nf)) continue;
                if (A[j].size() != A[i].size() || E[j].size() != E[i].size()) continue;
                if (!tagged) {
                    ++tcur;
                    for (int x : A[i]) tag[x] = tcur;
                    for (int x : E[i]) tag[x] = tcur;
                    tagged = true;
                }
                bool same = true;
                for (int x : A[j]) if (tag[x] != tcur) { same = false; break; }
                if (same) for (int x : E[j]) if (tag[x] != tcur) { same = false; break; }
                if (!same) continue;
                // j joins i
                member[last_member[i]] = j;
                last_member[i] = last_member[j];
                nv[i] += nv[j];
                nv[j] = 0;
                state[j] = DEAD;
                std::vector<int>().swap(A[j]);
                std::vector<int>().swap(E[j]);
            }
        }
        // ---- the element's variable list, final degrees, back into the buckets
        std::vector<int>& Lnew = L[p];
        Lnew.clear();
        for (int i : Lp)
            if (state[i] == VAR) Lnew.push_back(i);
        Lsize[p] = degLp;
        for (int i : Lnew) {
            int64_t d = (int64_t)deg[i] + degLp - nv[i];
            d = std::min<int64_t>(d, (int64_t)nleft - nv[i]);
            if (d < 0) d = 0;
            deg[i] = (int)d;
            if (i < nf) {
                bucket_insert(i);
                if (deg[i] < mindeg) mindeg = deg[i];
            }
        }
        nv[p] = nvp;
        if (Lnew.empty()) state[p] = DEAD;
        emit_members(p);
    }
    for (int v : nodes) local[v] = -1;
    for (int v : halo) local[v] = -1;
}

// =====================================================================================================
// nested dissection
// =====================================================================================================
namespace {

// ---------------------------------------------------------------------------------------------------
// Multilevel edge bisection (heavy-edge matching, graph-growing initial partitions, Fiduccia-Mattheyses refinement
// while uncoarsening -- the scheme of Hendrickson & Leland 1995 / Karypis & Kumar 1998, written from the papers).
// Used to find the two sides of a dissection step; the vertex separator is then cut out of the boundary and refined
// by Dissector::refine.
// ---------------------------------------------------------------------------------------------------
struct CGraph {
    int n = 0;
    std::vector<int> xadj, adj, ew, vw;
    int64_t tvw = 0;
    int maxvw = 1;
};

inline uint32_t next_rand(uint64_t& s) {
    s = s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (uint32_t)(s >> 33);
}

void coarsen_once(const CGraph& g, CGraph& c, std::vector<int>& cmap, int target, uint64_t& rng) {
    const int n = g.n;
    std::vector<int> match(n, -1), perm(n), rep;
    std::iota(perm.begin(), perm.end(), 0);
    for (int i = n - 1; i > 0; --i) std::swap(perm[i], perm[next_rand(rng) % (uint32_t)(i + 1)]);
    const int64_t maxvw = std::max<int64_t>(1, (3 * g.tvw) / (2 * std::max(1, target)));
    cmap.assign(n, -1);
    int nc = 0;
    for (int v : perm) {
        if (match[v] >= 0) continue;
        int best = -1, bw = -1;
        for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
            const int u = g.adj[k];
            if (match[u] < 0 && u != v && (int64_t)g.vw[v] + g.vw[u] <= maxvw && g.ew[k] > bw) { best = u; bw = g.ew[k]; }
        }
        if (best >= 0) { match[v] = best; match[best] = v; cmap[best] = nc; }
        else match[v] = v;
        cmap[v] = nc++;
        rep.push_back(v);
    }
    c = CGraph();
    c.n = nc;
    c.tvw = g.tvw;
    c.vw.assign(nc, 0);
    c.xadj.assign(nc + 1, 0);
    std::vector<int> pos(nc, -1);
    for (int cv = 0; cv < nc; ++cv) {
        const int start = (int)c.adj.size();
        const int v = rep[cv], u = match[v];
        for (int pass = 0; pass < (u == v ? 1 : 2); ++pass) {
            const int x = pass ? u : v;
            c.vw[cv] += g.vw[x];
            for (int k = g.xadj[x]; k < g.xadj[x + 1]; ++k) {
                const int cu = cmap[g.adj[k]];
                if (cu == cv) continue;
                if (pos[cu] < start) { pos[cu] = (int)c.adj.size(); c.adj.push_back(cu); c.ew.push_back(g.ew[k]); }
                else c.ew[pos[cu]] += g.ew[k];
            }
        }
        c.maxvw = std::max(c.maxvw, c.vw[cv]);
        c.xadj[cv + 1] = (int)c.adj.size();
    }
}

// FM refinement of a 2-way partition; returns the cut.  Vertices always leave the heavier side, best-gain first, through
// negative gains up to a look-ahead limit; the pass is rolled back to its best prefix.
int64_t fm_refine(const CGraph& g, std::vector<signed char>& where) {
    const int n = g.n;
    std::vector<int> id(n, 0), ed(n, 0), ver(n, 0), locked(n, -1);
    int64_t pw[2] = {0, 0}, cut = 0;
    for (int v = 0; v < n; ++v) {
        pw[where[v]] += g.vw[v];
        for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) (where[g.adj[k]] == where[v] ? id[v] : ed[v]) += g.ew[k];
        cut += ed[v];
    }
    cut /= 2;
    struct Entry { int gain, v, ver; };
    auto less = [](const Entry& a, const Entry& b) { return a.gain < b.gain; };
    const int64_t slack = std::max<int64_t>(g.maxvw, (int64_t)(0.03 * (double)g.tvw));
    auto balanced = [&](int64_t a, int64_t b) { return std::llabs(a - b) <= 2 * slack; };
    const int limit = std::min(std::max((int)(0.01 * n), 25), 150);
    for (int pass = 0; pass < 8; ++pass) {
        std::vector<Entry> heap[2];
        for (int v = 0; v < n; ++v)
            if (ed[v] > 0 || id[v] == 0) { heap[where[v]].push_back({ed[v] - id[v], v, ver[v]}); }
        for (int s = 0; s < 2; ++s) std::make_heap(heap[s].begin(), heap[s].end(), less);
        std::vector<int> log;
        int64_t best_cut = cut, best_imb = std::llabs(pw[0] - pw[1]);
        size_t best_len = 0;
        auto top_valid = [&](int s) -> bool {
            while (!heap[s].empty()) {
                const Entry& e = heap[s].front();
                if (where[e.v] == s && e.ver == ver[e.v] && locked[e.v] != pass) return true;
                std::pop_heap(heap[s].begin(), heap[s].end(), less);
                heap[s].pop_back();
            }
            return false;
        };
        while (true) {
            const bool h0 = top_valid(0), h1 = top_valid(1);
            if (!h0 && !h1) break;
            int from;
            if (!balanced(pw[0], pw[1])) from = pw[0] > pw[1] ? 0 : 1;
            else if (h0 && h1) from = heap[0].front().gain >= heap[1].front().gain ? 0 : 1;
            else from = h0 ? 0 : 1;
            if (!(from ? h1 : h0)) break;
            std::pop_heap(heap[from].begin(), heap[from].end(), less);
            const Entry e = heap[from].back();
            heap[from].pop_back();
            const int v = e.v, to = 1 - from;
            if (balanced(pw[0], pw[1]) && !balanced(pw[from] - g.vw[v], pw[to] + g.vw[v]) && pw[to] + g.vw[v] > pw[from] - g.vw[v]) {
                locked[v] = pass;        // would unbalance: skip this vertex for the pass
                continue;
            }
            cut -= ed[v] - id[v];
            std::swap(id[v], ed[v]);
            where[v] = (signed char)to;
            pw[from] -= g.vw[v];
            pw[to] += g.vw[v];
            locked[v] = pass;
            for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
                const int u = g.adj[k], w = g.ew[k];
                if (where[u] == to) { id[u] += w; ed[u] -= w; }
                else { id[u] -= w; ed[u] += w; }
                if (locked[u] != pass) {
                    ++ver[u];
                    heap[where[u]].push_back({ed[u] - id[u], u, ver[u]});
                    std::push_heap(heap[where[u]].begin(), heap[where[u]].end(), less);
                }
            }
            log.push_back(v);
            const int64_t imb = std::llabs(pw[0] - pw[1]);
            const bool ok = balanced(pw[0], pw[1]) || imb < best_imb;
            if (ok && (cut < best_cut || (cut == best_cut && imb < best_imb))) {
                best_cut = cut;
                best_imb = imb;
                best_len = log.size();
            } else if ((int)(log.size() - best_len) > limit) break;
        }
        while (log.size() > best_len) {
            const int v = log.back();
            log.pop_back();
            const int from = where[v], to = 1 - from;
            cut -= ed[v] - id[v];
            std::swap(id[v], ed[v]);
            where[v] = (signed char)to;
            pw[from] -= g.vw[v];
            pw[to] += g.vw[v];
            for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k) {
                const int u = g.adj[k], w = g.ew[k];
                if (where[u] == to) { id[u] += w; ed[u] -= w; }
                else { id[u] -= w; ed[u] += w; }
            }
        }
        for (int v : log) ++ver[v];
        if (best_len == 0) break;
    }
    return cut;
}

// graph-growing initial bisections of the coarsest graph, each FM-refined; the best cut wins
void initial_bisection(const CGraph& g, std::vector<signed char>& where, uint64_t& rng) {
    const int n = g.n;
    int64_t best = -1;
    std::vector<signed char> w(n);
    std::vector<int> queue;
    std::vector<char> seen(n);
    for (int trial = 0; trial < 10; ++trial) {
        std::fill(w.begin(), w.end(), (signed char)1);
        std::fill(seen.begin(), seen.end(), 0);
        int64_t pw0 = 0;
        queue.clear();
        size_t qh = 0;
        int nseen = 0;
        int seed = (int)(next_rand(rng) % (uint32_t)n);
        while (2 * pw0 < g.tvw) {
            if (qh == queue.size()) {    // start, or the component is exhausted
                if (nseen == n) break;
                while (seen[seed]) seed = (seed + 1) % n;
                seen[seed] = 1;
                ++nseen;
                queue.push_back(seed);
            }
            const int v = queue[qh++];
            if (2 * pw0 + g.vw[v] > g.tvw) continue;               // overshoots by more than it helps
            w[v] = 0;
            pw0 += g.vw[v];
            for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
                if (!seen[g.adj[k]]) { seen[g.adj[k]] = 1; ++nseen; queue.push_back(g.adj[k]); }
        }
        const int64_t cut = fm_refine(g, w);
        if (best < 0 || cut < best) { best = cut; where = w; }
    }
}

// two sides of the graph: where[v] in {0, 1}
void multilevel_bisection(const CGraph& g0, std::vector<signed char>& where, uint64_t seed) {
    uint64_t rng = seed;
    std::vector<CGraph> levels;
    std::vector<std::vector<int>> cmaps;
    levels.push_back(g0);
    const int target = 100;
    while (levels.back().n > target) {
        CGraph c;
        std::vector<int> cmap;
        coarsen_once(levels.back(), c, cmap, target, rng);
        if (c.n > 0.95 * levels.back().n) break;       // matching stalled (star-like graphs)
        levels.push_back(std::move(c));
        cmaps.push_back(std::move(cmap));
    }
    std::vector<signed char> w;
    initial_bisection(levels.back(), w, rng);
    for (int l = (int)levels.size() - 2; l >= 0; --l) {
        std::vector<signed char> fine(levels[l].n);
        for (int v = 0; v < levels[l].n; ++v) fine[v] = w[cmaps[l][v]];
        fm_refine(levels[l], fine);
        w.swap(fine);
    }
    where.swap(w);
}

struct Dissector {
    const Graph& adj;
    std::unique_ptr<std::atomic<int>[]> part;     // relaxed atomics: sibling subproblems relabel their own nodes concurrently
    std::vector<int> level;
    std::vector<int>& order;                      // preallocated, every call fills its own range [off, off + |nodes|)
    std::atomic<int> next_id{1};
    int leaf_size;
    bool amd_leaves;
    int nd_mode = 3;                              // 1 level sets, 2 multilevel bisection, 3 level sets + multilevel where they are jagged, 7 always both
    bool norefine = false;
    std::function<void(size_t, size_t)> on_top;   // called with (separator size, part size) for every split at depth 0
    std::vector<signed char> where;               // separator refinement: 0 left, 1 right, 2 separator
    std::vector<int> lock_stamp, version;
    std::atomic<int> next_pass{1};
    Dissector(const Graph& a, std::vector<int>& o, int leaf, bool amd_in_leaves)
        : adj(a), part(new std::atomic<int>[a.size() ? a.size() : 1]), level(a.size(), -1), order(o), leaf_size(leaf),
          amd_leaves(amd_in_leaves),
          where(a.size(), 0), lock_stamp(a.size(), 0), version(a.size(), 0) {
        for (size_t i = 0; i < a.size(); ++i) part[i].store(0, std::memory_order_relaxed);
    }
    int part_of(int v) const { return part[v].load(std::memory_order_relaxed); }

    // BFS inside part `id` from root; fills levels, leaves level[] set for the visited nodes
    void bfs(int root, int id, std::vector<std::vector<int>>& levels) {
        levels.clear();
        std::vector<int> cur{root};
        level[root] = 0;
        while (!cur.empty()) {
            levels.push_back(cur);
            std::vector<int> nxt;
            for (int v : cur)
                for (int u : adj[v])
                    if (part_of(u) == id && level[u] < 0) {
                        level[u] = (int)levels.size();
                        nxt.push_back(u);
                    }
            cur.swap(nxt);
        }
    }
    void clear(const std::vector<std::vector<int>>& levels) {
        for (auto& L : levels)
            for (int v : L) level[v] = -1;
    }

    // the two sides from the multilevel edge bisection of the part; the separator is the boundary of the side with the
    // smaller boundary (refine() then shrinks it)
    void multilevel_sides(const std::vector<int>& nodes, int id, std::vector<int>& left, std::vector<int>& right,
                          std::vector<int>& sep) {
        thread_local std::vector<int> lmap;
        if (lmap.size() != adj.size()) lmap.assign(adj.size(), -1);
        const int nn = (int)nodes.size();
        for (int k = 0; k < nn; ++k) lmap[nodes[k]] = k;
        CGraph g;
        g.n = nn;
        g.tvw = nn;
        g.vw.assign(nn, 1);
        g.xadj.assign(nn + 1, 0);
        for (int k = 0; k < nn; ++k) {
            for (int u : adj[nodes[k]])
                if (lmap[u] >= 0 && part_of(u) == id) { g.adj.push_back(lmap[u]); g.ew.push_back(1); }
            g.xadj[k + 1] = (int)g.adj.size();
        }
        std::vector<signed char> w;
        multilevel_bisection(g, w, 0x9E3779B97F4A7C15ULL ^ (uint64_t)nn);
        int nb[2] = {0, 0};
        std::vector<char> boundary(nn, 0);
        for (int v = 0; v < nn; ++v)
            for (int k = g.xadj[v]; k < g.xadj[v + 1]; ++k)
                if (w[g.adj[k]] != w[v]) { boundary[v] = 1; nb[w[v]]++; break; }
        const int sside = nb[0] <= nb[1] ? 0 : 1;
        for (int v = 0; v < nn; ++v) {
            if (boundary[v] && w[v] == sside) sep.push_back(nodes[v]);
            else (w[v] == 0 ? left : right).push_back(nodes[v]);
        }
        for (int k = 0; k < nn; ++k) lmap[nodes[k]] = -1;
    }

    // Vertex-separator refinement (Fiduccia-Mattheyses moves on the separator, as in multilevel nested dissection codes):
    // moving a separator node v to side `to` pulls its neighbours on the other side into the separator, gain =
    // 1 - (number pulled).  Moves are taken best-gain first, also through zero / negative gains (bounded look-ahead), and
    // the sequence is rolled back to the smallest separator seen.  No edge ever joins the two sides.
    void refine(int id, std::vector<int>& left, std::vector<int>& right, std::vector<int>& sep) {
        for (int v : left) where[v] = 0;
        for (int v : right) where[v] = 1;
        for (int v : sep) where[v] = 2;
        int64_t pw[3] = {(int64_t)left.size(), (int64_t)right.size(), (int64_t)sep.size()};
        const int64_t total = pw[0] + pw[1] + pw[2];
        const int64_t maxside = (int64_t)(0.6 * (double)total) + 1;
        struct Entry { int gain; int64_t tie; int v, to, ver; };
        auto less = [](const Entry& a, const Entry& b) { return a.gain != b.gain ? a.gain < b.gain : a.tie < b.tie; };
        struct Move { int v, to; std::vector<int> pulled; };
        auto count_side = [&](int v, int side) {
            int c = 0;
            for (int u : adj[v])
                if (part_of(u) == id && where[u] == side) ++c;
            return c;
        };
        for (int pass = 0; pass < 6; ++pass) {
            const int stamp = next_pass.fetch_add(1);
            std::vector<Entry> heap;
            auto push_both = [&](int x) {
                ++version[x];
                for (int to = 0; to < 2; ++to) {
                    heap.push_back({1 - count_side(x, 1 - to), -pw[to], x, to, version[x]});
                    std::push_heap(heap.begin(), heap.end(), less);
                }
            };
            std::vector<int> cur_sep;
            for (int v : left) if (where[v] == 2) cur_sep.push_back(v);
            for (int v : right) if (where[v] == 2) cur_sep.push_back(v);
            for (int v : sep) if (where[v] == 2) cur_sep.push_back(v);
            for (int v : cur_sep) push_both(v);
            std::vector<Move> log;
            int64_t best = pw[2], best_imb = std::llabs(pw[0] - pw[1]);
            size_t best_len = 0;
            const size_t patience = std::max<size_t>(64, cur_sep.size() / 4);
            while (!heap.empty() && log.size() - best_len < patience) {
                std::pop_heap(heap.begin(), heap.end(), less);
                const Entry e = heap.back();
                heap.pop_back();
                if (where[e.v] != 2 || e.ver != version[e.v] || lock_stamp[e.v] == stamp) continue;
                const int to = e.to, other = 1 - to;
                Move mv{e.v, to, {}};
                for (int u : adj[e.v])
                    if (part_of(u) == id && where[u] == other) mv.pulled.push_back(u);
                if (pw[to] + 1 > maxside && pw[to] >= pw[other]) continue;      // would only worsen a full side
                if (pw[other] - (int64_t)mv.pulled.size() < total / 5) continue; // keep both sides substantial
                where[e.v] = (signed char)to;
                lock_stamp[e.v] = stamp;
                pw[to]++;
                pw[2]--;
                for (int u : mv.pulled) {
                    where[u] = 2;
                    pw[other]--;
                    pw[2]++;
                }
                // gains change for the separator nodes around v and around the pulled nodes, and for the pulled nodes
                for (int x : adj[e.v])
                    if (part_of(x) == id && where[x] == 2 && lock_stamp[x] != stamp) push_both(x);
                for (int u : mv.pulled) {
                    if (lock_stamp[u] != stamp) push_both(u);
                    for (int x : adj[u])
                        if (part_of(x) == id && where[x] == 2 && lock_stamp[x] != stamp && x != u) push_both(x);
                }
                log.push_back(std::move(mv));
                const int64_t imb = std::llabs(pw[0] - pw[1]);
                if (pw[2] < best || (pw[2] == best && imb < best_imb)) {
                    best = pw[2];
                    best_imb = imb;
                    best_len = log.size();
                }
            }
            const bool improved = best_len > 0;
            while (log.size() > best_len) {     // roll back to the best prefix
                const Move& mv = log.back();
                const int other = 1 - mv.to;
                for (int u : mv.pulled) { where[u] = (signed char)other; pw[other]++; pw[2]--; }
                where[mv.v] = 2;
                pw[mv.to]--;
                pw[2]++;
                log.pop_back();
            }
            if (!improved) break;
        }
        std::vector<int> all;
        all.reserve((size_t)total);
        all.insert(all.end(), left.begin(), left.end());
        all.insert(all.end(), right.begin(), right.end());
        all.insert(all.end(), sep.begin(), sep.end());
        left.clear();
        right.clear();
        sep.clear();
        for (int v : all) (where[v] == 0 ? left : (where[v] == 1 ? right : sep)).push_back(v);
    }
    // a part that is not cut any further: constrained minimum degree (its neighbours outside the part are all in
    // separators that are ordered later, so they form the halo)
    void leaf(const std::vector<int>& nodes, int id, size_t off) {
        if (nodes.size() <= 3 || !amd_leaves) {
            for (size_t k = 0; k < nodes.size(); ++k) order[off + k] = nodes[k];
            return;
        }
        // halo: distinct neighbours outside the part.  part ids of finished separators are negative markers (-1)
        std::vector<int> halo;
        for (int v : nodes)
            for (int u : adj[v])
                if (part_of(u) != id) halo.push_back(u);
        std::sort(halo.begin(), halo.end());
        halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
        thread_local std::vector<int> local;      // amd_order's n-sized -1 map, reused across the leaves of a thread
        if (local.size() != adj.size()) local.assign(adj.size(), -1);
        amd_order(adj, nodes, halo, local, order.data() + off);
    }
    void run(std::vector<int>& nodes, int depth, size_t off) {
        if (nodes.empty()) return;
        const int id = next_id.fetch_add(1);
        for (int v : nodes) part[v].store(id, std::memory_order_relaxed);
        std::vector<std::vector<int>> levels;
        // first sweep from nodes[0]: if it does not reach every node the part is disconnected -> one call per component
        bfs(nodes[0], id, levels);
        size_t reached = 0;
        for (auto& L : levels) reached += L.size();
        if (reached != nodes.size()) {
            std::vector<std::vector<int>> comps;
            {
                std::vector<int> c;
                for (auto& L : levels) c.insert(c.end(), L.begin(), L.end());
                comps.push_back(std::move(c));
            }
            for (int v : nodes)
                if (level[v] < 0) {
                    bfs(v, id, levels);
                    std::vector<int> c;
                    for (auto& L : levels) c.insert(c.end(), L.begin(), L.end());
                    comps.push_back(std::move(c));
                }
            for (int v : nodes) level[v] = -1;
            for (auto& c : comps) {
                run(c, depth, off);
                off += c.size();
            }
            return;
        }
        if ((int)nodes.size() <= leaf_size || depth > 60) {
            clear(levels);
            leaf(nodes, id, off);
            return;
        }
        // pseudo-peripheral root: restart from a minimum-degree node of the last level, at most twice more; the level
        // structure of the last sweep is the one that is cut
        int root = nodes[0];
        for (int sweep = 0; sweep < 3; ++sweep) {
            int best = levels.back()[0];
            for (int v : levels.back())
                if (adj[v].size() < adj[best].size()) best = v;
            clear(levels);
            if (best == root || sweep == 2) break;
            root = best;
            bfs(root, id, levels);
        }
        if (levels.size() < 3) {   // (nearly) complete graph: no useful separator
            leaf(nodes, id, off);
            return;
        }
        const size_t total = nodes.size();
        std::vector<size_t> prefix(levels.size() + 1, 0);
        for (size_t l = 0; l < levels.size(); ++l) prefix[l + 1] = prefix[l] + levels[l].size();
        size_t best_l = levels.size() / 2;
        double best_cost = 1e300;
        for (size_t l = 1; l + 1 < levels.size(); ++l) {
            const double a = (double)prefix[l], b = (double)(total - prefix[l + 1]);
            const double imbalance = std::abs(a - b) / (double)total;
            const double cost = (double)levels[l].size() * (1.0 + 4.0 * imbalance);
            if (imbalance < 0.6 && cost < best_cost) {
                best_cost = cost;
                best_l = l;
            }
        }
        // thin the level to a minimal separator: a node of level l without a neighbour in level l+1 separates nothing
        // and joins the left side (levels < l); level[] was cleared, so recompute membership with a local stamp
        std::vector<int> left, right, sep;
        for (size_t l = 0; l < best_l; ++l) left.insert(left.end(), levels[l].begin(), levels[l].end());
        for (size_t l = best_l + 1; l < levels.size(); ++l) right.insert(right.end(), levels[l].begin(), levels[l].end());
        for (int v : levels[best_l + 1]) level[v] = 1;
        for (int v : levels[best_l]) {
            bool needed = false;
            for (int u : adj[v])
                if (level[u] == 1 && part_of(u) == id) { needed = true; break; }
            (needed ? sep : left).push_back(v);
        }
        for (int v : levels[best_l + 1]) level[v] = -1;
        const size_t sep0 = sep.size();
        const size_t raw = levels[best_l].size();
        if (!norefine && (nd_mode & 1)) refine(id, left, right, sep);
        // the multilevel bisection is the expensive candidate: it pays when the level set was a poor separator, which
        // shows as a large reduction by the thinning + refinement (jagged level sets of unstructured meshes); the level
        // sets of regular grids come out of the refinement unchanged
        const bool jagged = 10 * sep.size() < 8 * raw && nodes.size() >= 600;    // small parts: their separators carry no weight
        if ((nd_mode & 2) && (jagged || nd_mode == 2 || nd_mode == 7)) {
            std::vector<int> l2, r2, s2;
            multilevel_sides(nodes, id, l2, r2, s2);
            if (!s2.empty() && !l2.empty() && !r2.empty()) {
                if (!norefine) refine(id, l2, r2, s2);
                auto cost = [&](const std::vector<int>& a, const std::vector<int>& b, const std::vector<int>& c) {
                    const double imb = std::abs((double)a.size() - (double)b.size()) / (double)total;
                    return (double)c.size() * (1.0 + 2.0 * imb);
                };
                if (nd_mode == 2 || cost(l2, r2, s2) < cost(left, right, sep)) {
                    left.swap(l2);
                    right.swap(r2);
                    sep.swap(s2);
                }
            }
        }
        for (int v : sep) part[v].store(-1, std::memory_order_relaxed);
        if (depth == 0 && on_top) on_top(sep.size(), nodes.size());
        if (depth < 3 && dev_knob("MI355KKT_ND_DEBUG"))
            fprintf(stderr, "[nd] depth %d: %zu nodes -> left %zu right %zu separator %zu (level %zu of %zu, raw %zu, thinned %zu)\n", depth, nodes.size(),
                    left.size(), right.size(), sep.size(), best_l, levels.size(), levels[best_l].size(), sep0);
        const size_t off_r = off + left.size(), off_s = off_r + right.size();
        // the two sides are independent subproblems (disjoint nodes, disjoint output ranges): the first few levels of
        // the recursion run them on separate host threads
        if (depth < 5 && left.size() > 2048 && right.size() > 2048) {
            std::future<void> fut;
            try {
                fut = std::async(std::launch::async, [&]() { run(left, depth + 1, off); });
            } catch (const std::system_error&) {   // no thread to be had: same work on this one
                run(left, depth + 1, off);
            }
            run(right, depth + 1, off_r);
            if (fut.valid()) fut.get();
        } else {
            run(left, depth + 1, off);
            run(right, depth + 1, off_r);
        }
        for (size_t k = 0; k < sep.size(); ++k) order[off_s + k] = sep[k];
    }
};

}  // namespace

void fill_reducing_ordering(const Graph& adj, std::vector<int>& order, int method, OrderingInfo* info) {
    const int n = (int)adj.size();
    OrderingInfo I;
    order.assign(n, -1);
    if (n == 0) { if (info) *info = I; return; }
    if (const char* e = dev_knob("MI355KKT_ORDERING")) {
        if (!strcmp(e, "nd")) method = 1;
        else if (!strcmp(e, "amd")) method = 2;
        else if (!strcmp(e, "auto")) method = 0;
    }
    // leaves of the dissection: breadth-first order (chains that amalgamate into few, wide supernodes: what the small-front
    // kernels like) or, with MI355KKT_ND_LEAF_AMD, constrained minimum degree (about 6 % less fill, 3x more supernodes)
    const bool amd_leaves = dev_knob("MI355KKT_ND_LEAF_AMD") != nullptr;
    const int leaf = dev_knob("MI355KKT_ND_LEAF") ? atoi(dev_knob("MI355KKT_ND_LEAF")) : (amd_leaves ? 120 : 48);
    const bool dbg = dev_knob("MI355KKT_SPARSE_DEBUG") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sparse]   %-26s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<int> ond, oamd;
    std::vector<int64_t> cc_nd, cc_amd;
    std::vector<int> par_nd, par_amd;
    auto run_amd = [&]() {
        // dense rows (degree > 10 sqrt(n)) would dominate the quotient-graph work and gain nothing: ordered last
        const size_t dense = (size_t)std::max(16.0, 10.0 * std::sqrt((double)n));
        std::vector<int> nodes, late;
        for (int v = 0; v < n; ++v) (adj[v].size() > dense ? late : nodes).push_back(v);
        std::stable_sort(late.begin(), late.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
        oamd.assign(n, -1);
        std::vector<int> local(n, -1);
        amd_order(adj, nodes, {}, local, oamd.data());
        std::copy(late.begin(), late.end(), oamd.begin() + nodes.size());
        etree_and_counts(adj, oamd, par_amd, cc_amd);
        cost_of(cc_amd, I.nnz_amd, I.flops_amd);
    };
    // The minimum-degree candidate runs beside the dissection, on its own thread -- but only when it can win.  It wins on
    // graphs without small separators (random / small-world graphs: top separator of 0.2 n nodes); on mesh-like graphs, whose
    // top separator is O(n^(2/3)) or smaller, the dissection is better in arithmetic AND in tree height in every case measured
    // (2-D / 3-D grids, Delaunay meshes, band matrices: cost ratio 1.35 .. 90), and the candidate cost 4/5 of the ordering time
    // (46^3: 130 ms beside the dissection's 28 ms).  So the candidate is started when the first split at depth 0 shows a
    // separator above 1.5 |part|^(2/3) -- a structural, deterministic test -- and always for small graphs.
    std::future<void> amd_future;
    bool amd_started = false, top_seen = false;
    auto start_amd = [&]() {
        if (amd_started) return;
        amd_started = true;
        try {
            amd_future = std::async(std::launch::async, run_amd);
        } catch (const std::system_error&) {
            run_amd();
        }
    };
    const bool amd_always = dev_knob("MI355KKT_ORDERING_BOTH") != nullptr;
    if (method == 0 && (n < 2000 || amd_always)) start_amd();
    else if (method == 2) run_amd();
    if (method != 2) {
        ond.assign(n, -1);
        std::vector<int> nodes(n);
        std::iota(nodes.begin(), nodes.end(), 0);
        Dissector nd(adj, ond, leaf, amd_leaves);
        if (const char* e = dev_knob("MI355KKT_ND_MODE")) nd.nd_mode = atoi(e);
        nd.norefine = dev_knob("MI355KKT_ND_NOREFINE") != nullptr;
        if (method == 0)
            nd.on_top = [&](size_t sep, size_t part) {
                top_seen = true;
                if ((double)sep > 1.5 * std::pow((double)part, 2.0 / 3.0)) start_amd();
            };
        nd.run(nodes, 0, 0);
        if (method == 0 && !top_seen) start_amd();        // nothing was split (dense / tiny parts): both candidates
        lap("dissection");
        etree_and_counts(adj, ond, par_nd, cc_nd);
        cost_of(cc_nd, I.nnz_nd, I.flops_nd);
        lap("its tree and counts");
    }
    if (amd_future.valid()) amd_future.get();
    lap("minimum degree (rest)");
    // relabel a candidate along the postorder of its elimination tree and measure the height of its supernodal tree
    auto finish = [&](const std::vector<int>& o, const std::vector<int>& par, const std::vector<int64_t>& cc,
                      std::vector<int>& out, int& height, std::vector<int>& par2, std::vector<int64_t>& cc2) {
        std::vector<int> post, newidx(n);
        par2.resize(n);
        cc2.resize(n);
        postorder(par, cc, post);
        for (int k = 0; k < n; ++k) newidx[post[k]] = k;
        out.resize(n);
        for (int k = 0; k < n; ++k) {
            out[k] = o[post[k]];
            par2[k] = par[post[k]] < 0 ? -1 : newidx[par[post[k]]];
            cc2[k] = cc[post[k]];
        }
        height = supernodal_height(par2, cc2);
    };
    std::vector<int> fnd, famd, p2nd, p2amd;
    std::vector<int64_t> c2nd, c2amd;
    if (method != 2) finish(ond, par_nd, cc_nd, fnd, I.levels_nd, p2nd, c2nd);
    const bool have_amd = method == 2 || (method == 0 && amd_started);
    if (have_amd) finish(oamd, par_amd, cc_amd, famd, I.levels_amd, p2amd, c2amd);
    // cost model of the device engine: the arithmetic at the rate the front kernels sustain + a fixed cost per level of
    // the supernodal tree (one factor step and two solve steps are launched per level)
    auto seconds = [](double flops, int levels) { return flops / 2.0e12 + 1.5e-4 * (double)levels; };
    bool use_amd = method == 2;
    if (method == 0 && have_amd) use_amd = seconds(I.flops_amd, I.levels_amd) < 0.9 * seconds(I.flops_nd, I.levels_nd);
    I.method = use_amd ? 2 : 1;
    order.swap(use_amd ? famd : fnd);
    I.parent.swap(use_amd ? p2amd : p2nd);
    I.colcount.swap(use_amd ? c2amd : c2nd);
    lap("postorder, heights, choice");
    if (info) *info = I;
}

}  // namespace mi355kkt
