// symbolic.cpp — host symbolic phase of the MI355X transient solver.
//
// Replaces what the reference redoes on every iteration (dense allocation simulateTRAN.ts:152-153,
// partial-pivot search and row swaps solveReal.ts:15-34) by a ONE-TIME analysis per topology:
//   1. MNA pattern from the element lists (stampAdmittanceReal.ts:3-29, stampVoltageSourceReal.ts:4-32)
//   2. row matching for a zero-free diagonal: voltage-source branch rows have A[j][j] = 0
//      (stampVoltageSourceReal.ts:26-31); the reference relies on partial pivoting, here the branch
//      equation is matched to one of its node columns (entry +-1) once, statically
//   3. nested-dissection ordering by BFS level-structure separators: short elimination tree
//      (log2 N levels on a ladder = cyclic reduction) so that each level is one parallel GPU phase
//   4. symbolic LU on the symmetrised pattern, elimination-tree levels
//   5. compilation into gather-form task lists (no atomics, deterministic summation order):
//      static/dynamic stamp lists, right-hand-side lists, per-level update tasks (factorisation with
//      the forward elimination fused in as an extra column), per-level backward-substitution tasks.
#include "symbolic.h"

#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <numeric>

namespace {

typedef std::vector<std::vector<int>> Adj;

void sort_unique(std::vector<int> &v) {
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
}

// ---------------------------------------------------------------------------------------------
// 2. maximum transversal (MC21-style augmenting DFS, iterative).  rows_of_col[c] lists candidate
// rows (preferred first).  Returns row_of_col[] or fails.
bool max_transversal(int n, const Adj &rows_of_col, std::vector<int> &row_of_col) {
  std::vector<int> col_of_row(n, -1);
  row_of_col.assign(n, -1);
  // cheap pass: diagonal first, then any free row
  for (int c = 0; c < n; c++)
    for (int r : rows_of_col[c])
      if (r == c && col_of_row[r] < 0) {
        row_of_col[c] = r;
        col_of_row[r] = c;
        break;
      }
  for (int c = 0; c < n; c++) {
    if (row_of_col[c] >= 0) continue;
    for (int r : rows_of_col[c])
      if (col_of_row[r] < 0) {
        row_of_col[c] = r;
        col_of_row[r] = c;
        break;
      }
  }
  std::vector<int> visited(n, -1), stack_c, stack_i;
  for (int c0 = 0; c0 < n; c0++) {
    if (row_of_col[c0] >= 0) continue;
    // DFS over columns; path stored in stack_c with iterator positions stack_i
    stack_c.assign(1, c0);
    stack_i.assign(1, 0);
    std::vector<int> path_row;  // row chosen at each depth
    bool found = false;
    while (!stack_c.empty()) {
      int c = stack_c.back();
      int &i = stack_i.back();
      if (i == 0) {
        // look for a free row first
        for (int r : rows_of_col[c])
          if (col_of_row[r] < 0) {
            path_row.push_back(r);
            found = true;
            break;
          }
        if (found) break;
      }
      bool descended = false;
      while (i < (int)rows_of_col[c].size()) {
        int r = rows_of_col[c][i++];
        if (visited[r] == c0) continue;
        visited[r] = c0;
        int c2 = col_of_row[r];
        if (c2 < 0) continue;  // (handled above)
        path_row.push_back(r);
        stack_c.push_back(c2);
        stack_i.push_back(0);
        descended = true;
        break;
      }
      if (!descended) {
        stack_c.pop_back();
        stack_i.pop_back();
        if (!path_row.empty() && !stack_c.empty()) path_row.pop_back();
      }
    }
    if (!found) return false;
    // augment along the path: stack_c[d] takes path_row[d]
    for (int d = (int)stack_c.size() - 1; d >= 0; d--) {
      int c = stack_c[d], r = path_row[d];
      row_of_col[c] = r;
      col_of_row[r] = c;
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// 3. nested dissection on an undirected graph.
struct NDOrder {
  const Adj &g;
  int n;
  std::vector<int> order;        // elimination order (list of vertices)
  std::vector<int> owner;        // current sub-problem id of each vertex (-1 = already ordered)
  std::vector<int> dist, queue_;
  int next_id = 1;

  explicit NDOrder(const Adj &g_) : g(g_), n((int)g_.size()), owner(g_.size(), 0), dist(g_.size(), -1) { order.reserve(n); }

  // BFS inside sub-problem `id` from `s`; fills queue_ (visit order) and dist; returns eccentricity
  int bfs(int s, int id) {
    queue_.clear();
    queue_.push_back(s);
    dist[s] = 0;
    size_t head = 0;
    while (head < queue_.size()) {
      int v = queue_[head++];
      for (int w : g[v])
        if (owner[w] == id && dist[w] < 0) {
          dist[w] = dist[v] + 1;
          queue_.push_back(w);
        }
    }
    return dist[queue_.back()];
  }
  void clear_dist() {
    for (int v : queue_) dist[v] = -1;
  }
  int sub_degree(int v, int id) const {
    int d = 0;
    for (int w : g[v]) d += owner[w] == id;
    return d;
  }

  void run() {
    // iterative over a work stack of sub-problems (vertex lists)
    std::vector<std::vector<int>> work;
    {
      std::vector<int> all;  // vertices the caller has already ordered (owner -1) stay out
      for (int v = 0; v < n; v++)
        if (owner[v] != -1) all.push_back(v);
      work.push_back(all);
    }
    // Sub-problems must be ordered so that separators come AFTER both halves.  Use a recursive
    // lambda with an explicit depth guard instead of a work stack to keep that order simple.
    std::function<void(std::vector<int> &, int)> rec = [&](std::vector<int> &verts, int depth) {
      if (verts.empty()) return;
      int id = next_id++;
      for (int v : verts) owner[v] = id;
      // split into connected components
      std::vector<std::vector<int>> comps;
      for (int v : verts) {
        if (dist[v] >= 0) continue;
        bfs(v, id);
        comps.emplace_back(queue_);
        // keep dist marks until all components are found
        for (int w : queue_) dist[w] = 1 << 30;
      }
      for (int v : verts) dist[v] = -1;
      if (comps.size() > 1) {
        for (auto &c : comps) rec(c, depth + 1);
        return;
      }
      std::vector<int> &S = comps[0];
      if (S.size() <= 2 || depth > 200) {
        leaf(S, id);
        return;
      }
      // pseudo-peripheral start
      int s = S[0], sd = sub_degree(S[0], id);
      for (int v : S) {
        int dg = sub_degree(v, id);
        if (dg < sd) { sd = dg; s = v; }
      }
      int ecc = -1;
      for (int it = 0; it < 8; it++) {
        int e = bfs(s, id);
        // candidate: min-degree vertex of the last level
        int best = -1, bd = 1 << 30;
        for (int i = (int)queue_.size() - 1; i >= 0 && dist[queue_[i]] == e; i--) {
          int dg = sub_degree(queue_[i], id);
          if (dg < bd) { bd = dg; best = queue_[i]; }
        }
        clear_dist();
        if (e <= ecc) break;
        ecc = e;
        s = best;
      }
      int h = bfs(s, id);
      if (h < 2) {
        clear_dist();
        leaf(S, id);
        return;
      }
      // level sizes
      std::vector<int> lsize(h + 1, 0);
      for (int v : queue_) lsize[dist[v]]++;
      int total = (int)S.size();
      // choose the separator level: smallest level among those leaving >= 30 % on each side,
      // falling back to the level where the cumulative count crosses one half
      int m = -1, best_sz = 1 << 30, best_imb = 1 << 30, cum = 0, median = 1;
      for (int l = 0; l <= h; l++) {
        int below = cum, above = total - cum - lsize[l];
        if (l >= 1 && l <= h - 1) {
          const int imb = below > above ? below - above : above - below;
          if (below * 10 >= total * 3 && above * 10 >= total * 3 &&
              (lsize[l] < best_sz || (lsize[l] == best_sz && imb < best_imb))) {
            best_sz = lsize[l];
            best_imb = imb;
            m = l;
          }
          if (cum * 2 < total) median = l;
        }
        cum += lsize[l];
      }
      if (m < 0) m = std::min(std::max(median, 1), h - 1);
      std::vector<int> A, B, sep;
      for (int v : queue_) {
        int dv = dist[v];
        if (dv < m) A.push_back(v);
        else if (dv > m) B.push_back(v);
        else {
          bool touches = false;
          for (int w : g[v])
            if (owner[w] == id && dist[w] == m + 1) { touches = true; break; }
          (touches ? sep : A).push_back(v);
        }
      }
      clear_dist();
      for (int v : sep) owner[v] = -2;  // removed from both halves
      rec(A, depth + 1);
      rec(B, depth + 1);
      for (int v : sep) {
        order.push_back(v);
        owner[v] = -1;
      }
    };
    rec(work[0], 0);
  }

  void leaf(std::vector<int> &S, int id) {
    // small or unsplittable blob: ascending sub-degree, ties by vertex number
    std::vector<std::pair<int, int>> key;
    for (int v : S) key.emplace_back(sub_degree(v, id), v);
    std::sort(key.begin(), key.end());
    for (auto &kv : key) {
      order.push_back(kv.second);
      owner[kv.second] = -1;
    }
  }
};

struct EntryIndex {
  // CSR of the permuted L+U pattern.  Entry IDS (the W indices the device uses) are a renumbering of the
  // CSR positions by class: [dynamic | static update targets | static never-modified], so that only the first
  // two classes have to be re-stamped every step and the dynamic ones are contiguous (wave-uniform paths).
  std::vector<int> ptr, col, id_of_pos, row_of_id, col_of_id;
  int pos(int r, int c) const {
    auto b = col.begin() + ptr[r], e = col.begin() + ptr[r + 1];
    auto it = std::lower_bound(b, e, c);
    return (it != e && *it == c) ? (int)(it - col.begin()) : -1;
  }
  int find(int r, int c) const {
    const int p = pos(r, c);
    return p < 0 ? -1 : id_of_pos[p];
  }
};

template <class T>
size_t add_section(std::vector<uint8_t> &blob, std::vector<size_t> &offs, const std::vector<T> &v) {
  size_t off = (blob.size() + 15) & ~size_t(15);
  blob.resize(off + std::max<size_t>(v.size() * sizeof(T), 16));
  if (!v.empty()) memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
  offs.push_back(off);
  return off;
}

// Pack variable-length tasks into wave-sized slices (tasks sorted by descending length so that a
// slice's lanes have similar trip counts).  W = number of index words per product.
template <int W>
void pack_slices(std::vector<std::pair<uint32_t, std::vector<uint32_t>>> &tasks,  // (target, flat products)
                 std::vector<SpiceySlice> &slices, std::vector<uint32_t> &tgt, std::vector<uint32_t> &cnt,
                 std::vector<uint32_t> &pairs, std::vector<uint32_t> *aux_in = nullptr, std::vector<uint32_t> *aux_out = nullptr) {
  std::vector<int> idx(tasks.size());
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return tasks[a].second.size() > tasks[b].second.size(); });
  for (size_t s0 = 0; s0 < idx.size(); s0 += 64) {
    size_t s1 = std::min(idx.size(), s0 + 64);
    uint32_t len = (uint32_t)(tasks[idx[s0]].second.size() / W);
    SpiceySlice sl;
    sl.off = (uint32_t)pairs.size();
    sl.len = len;
    slices.push_back(sl);
    pairs.resize(pairs.size() + (size_t)len * W * 64, 0u);
    for (size_t l = 0; l < 64; l++) {
      if (s0 + l < s1) {
        auto &t = tasks[idx[s0 + l]];
        tgt.push_back(t.first);
        uint32_t c = (uint32_t)(t.second.size() / W);
        cnt.push_back(c);
        if (aux_in) aux_out->push_back((*aux_in)[idx[s0 + l]]);
        for (uint32_t j = 0; j < c; j++)
          for (int w = 0; w < W; w++) pairs[sl.off + ((size_t)j * W + w) * 64 + l] = t.second[(size_t)j * W + w];
      } else {
        tgt.push_back(SPICEY_TGT_PAD);
        cnt.push_back(0);
        if (aux_in) aux_out->push_back(0);
      }
    }
  }
}

}  // namespace

int64_t spicey_algorithmic_bytes(const SpiceyDesc *d, int32_t nnzA, int32_t nnzLU) {
  // B = 8(3 nnzA + 2 nnzLU) + 4(nnzA + nnzLU) + 8*4*Nvar + 16(E_C+E_L+E_D) + 8(nNodes + E_total)
  int64_t nvar = (int64_t)d->n_nodes + d->nV;
  int64_t etot = (int64_t)d->nR + d->nC + d->nL + d->nV + d->nS + d->nD;
  return 8 * (3 * (int64_t)nnzA + 2 * (int64_t)nnzLU) + 4 * ((int64_t)nnzA + nnzLU) + 32 * nvar +
         16 * ((int64_t)d->nC + d->nL + d->nD) + 8 * ((int64_t)d->n_nodes + etot);
}

static int32_t build_program_impl(const SpiceyDesc *d, HostProgram &hp, std::string &err, bool slot_major, int front_cut, bool pcr_top, bool hybrid = false);


// LDS cycles the operand reads of the compact records cost per solve (every half-wave group and operand role: the
// largest number of distinct addresses on one bank) and the conflict-free minimum (one per group and role).
void spicey_bank_cost(const HostProgram &hp, int64_t *cycles, int64_t *ideal) {
  *cycles = 0; *ideal = 0;
  const int nL = hp.hdr.nLevels;
  for (int p = 0; p < (int)hp.ph_cnt.size(); p++) {
    const bool ktask = p >= nL;
    for (uint32_t g0 = 0; g0 < hp.ph_cnt[p]; g0 += 32) {
      const uint32_t g1 = std::min(hp.ph_cnt[p], g0 + 32);
      for (int role = 0; role < 7; role++) {
        std::vector<std::vector<uint32_t>> bank(32);
        bool any = false;
        for (uint32_t t = g0; t < g1; t++) {
          const uint32_t *r = &hp.rec16[((size_t)hp.ph_first[p] + t) * 4];
          const uint32_t cnt = (r[0] >> 16) & 0xffu;
          if (cnt > 2) continue;
          const uint32_t f[7] = {r[0] & 0xffffu, r[1] & 0xffffu, r[1] >> 16, r[2] & 0xffffu, r[2] >> 16, r[3] & 0xffffu, r[3] >> 16};
          const int nf = ktask ? (cnt == 0 ? 2 : cnt == 1 ? 4 : 6) : (cnt == 0 ? 1 : cnt == 1 ? 4 : 7);
          if (role >= nf) continue;
          auto &b = bank[f[role] & 31];
          if (std::find(b.begin(), b.end(), f[role]) == b.end()) b.push_back(f[role]);
          any = true;
        }
        if (!any) continue;
        size_t worst = 1;
        for (auto &b : bank) worst = std::max(worst, b.size());
        *cycles += (int64_t)worst;
        *ideal += 1;
      }
    }
  }
}

// Two candidate numberings of the L+U entries (see build_program_impl): slot-major ("bank-aware") and plain CSR
// order.  For programs that run from LDS (16-bit records) both are compiled and the one whose operand reads cost fewer
// LDS cycles is kept (chains: 2.65 -> 1.88 conflict factor; small meshes are sometimes better off in CSR order).
// Circuits on the global-workspace path keep the CSR order: LDS banks do not matter there.
int32_t spicey_build_program(const SpiceyDesc *d, HostProgram &hp, std::string &err, bool bank_aware, int front_cut, bool pcr_top, bool hybrid) {
  hp = HostProgram();
  if (hybrid) return build_program_impl(d, hp, err, true, 0, pcr_top, true);  // (the leaf-owned id ranges need the slot-major numbering)
  int32_t rc = build_program_impl(d, hp, err, false, front_cut, pcr_top);
  if (rc != SPICEY_OK || hp.structurally_singular || !hp.hdr.has16 || !bank_aware) return rc;
  HostProgram alt;
  std::string err2;
  if (build_program_impl(d, alt, err2, true, 0, pcr_top) == SPICEY_OK && alt.hdr.has16) {
    int64_t c0, i0, c1, i1;
    spicey_bank_cost(hp, &c0, &i0);
    spicey_bank_cost(alt, &c1, &i1);
    if (c1 < c0) hp = std::move(alt);
  }
  return rc;
}

static int32_t build_program_impl(const SpiceyDesc *d, HostProgram &hp, std::string &err, const bool slot_major, int front_cut, const bool pcr_top, const bool hybrid) {
  if (!d) { err = "null descriptor"; return SPICEY_ERR_BAD_DESC; }
  if (d->abi_version != SPICEY_ABI_VERSION) { err = "abi_version mismatch"; return SPICEY_ERR_BAD_DESC; }
  const int nN = d->n_nodes, nR = d->nR, nC = d->nC, nL = d->nL, nV = d->nV, nS = d->nS, nD = d->nD;
  if (nN < 0 || nR < 0 || nC < 0 || nL < 0 || nV < 0 || nS < 0 || nD < 0 || d->n_inst < 1) {
    err = "negative count or n_inst < 1";
    return SPICEY_ERR_BAD_DESC;
  }
  const int n = nN + nV;
  if (n == 0) { err = "empty circuit"; return SPICEY_ERR_BAD_DESC; }
  auto chk = [&](const int32_t *a, const int32_t *b, int cnt, const char *what) -> bool {
    if (cnt > 0 && (!a || !b)) { err = std::string("null node array for ") + what; return false; }
    for (int i = 0; i < cnt; i++)
      if (a[i] < 0 || a[i] > nN || b[i] < 0 || b[i] > nN) { err = std::string("node id out of range in ") + what; return false; }
    return true;
  };
  if (!chk(d->R_n1, d->R_n2, nR, "R") || !chk(d->C_n1, d->C_n2, nC, "C") || !chk(d->L_n1, d->L_n2, nL, "L") ||
      !chk(d->V_n1, d->V_n2, nV, "V") || !chk(d->S_n1, d->S_n2, nS, "S") || !chk(d->S_cp, d->S_cn, nS, "S control") ||
      !chk(d->D_np, d->D_nm, nD, "D"))
    return SPICEY_ERR_BAD_DESC;
  if ((nR && !d->R_val) || (nC && !d->C_val) || (nL && !d->L_val) || (nS && (!d->S_ron || !d->S_roff || !d->S_von || !d->S_voff)) ||
      (nD && (!d->D_is || !d->D_n))) {
    err = "null value array";
    return SPICEY_ERR_BAD_DESC;
  }
  const int nOut = (d->n_out > 0 && d->out_nodes) ? d->n_out : nN;
  if (d->n_out > 0 && d->out_nodes)
    for (int i = 0; i < nOut; i++)
      if (d->out_nodes[i] < 0 || d->out_nodes[i] > nN) { err = "out_nodes id out of range"; return SPICEY_ERR_BAD_DESC; }

  // ---- 1. pattern of A, by rows and by columns (original numbering) ---------------------------
  Adj arow(n), acol(n);
  auto add = [&](int r, int c) {
    if (r < 0 || c < 0) return;
    arow[r].push_back(c);
    acol[c].push_back(r);
  };
  auto two_terminal = [&](const int32_t *a, const int32_t *b, int cnt) {
    for (int i = 0; i < cnt; i++) {
      int i1 = a[i] - 1, i2 = b[i] - 1;
      add(i1, i1); add(i2, i2);
      if (i1 >= 0 && i2 >= 0) { add(i1, i2); add(i2, i1); }
    }
  };
  two_terminal(d->R_n1, d->R_n2, nR);
  two_terminal(d->C_n1, d->C_n2, nC);
  two_terminal(d->L_n1, d->L_n2, nL);
  two_terminal(d->S_n1, d->S_n2, nS);
  two_terminal(d->D_np, d->D_nm, nD);
  // rows of each column with the voltage-source incidence rows FIRST (numerically +-1: preferred pivots)
  Adj vrows(n);
  for (int k = 0; k < nV; k++) {
    int i1 = d->V_n1[k] - 1, i2 = d->V_n2[k] - 1, j = nN + k;
    if (i1 == i2) continue;  // +1-1 cancels numerically; leave the branch row empty -> singular
    add(i1, j); add(i2, j); add(j, i1); add(j, i2);
    if (i1 >= 0) { vrows[i1].push_back(j); vrows[j].push_back(i1); }
    if (i2 >= 0) { vrows[i2].push_back(j); vrows[j].push_back(i2); }
  }
  hp.nnzA = 0;
  for (int r = 0; r < n; r++) { sort_unique(arow[r]); hp.nnzA += (int)arow[r].size(); }
  Adj cand(n);
  for (int c = 0; c < n; c++) {
    sort_unique(acol[c]);
    std::vector<int> pref = vrows[c];
    sort_unique(pref);
    cand[c] = pref;
    for (int r : acol[c])
      if (!std::binary_search(pref.begin(), pref.end(), r)) cand[c].push_back(r);
  }

  hp.hdr = SpiceyProg{};
  hp.hdr.n = n; hp.hdr.nR = nR; hp.hdr.nC = nC; hp.hdr.nL = nL; hp.hdr.nV = nV; hp.hdr.nS = nS; hp.hdr.nD = nD;
  hp.hdr.nU = nC + nL + nV + nD + nV;  /* + one slot per source for the NEXT step's value (v2, written during the last backward phase) */ hp.hdr.nGdyn = nS + nD; hp.hdr.nGstat = nR + nC + nL + 1;
  hp.hdr.nOut = nOut; hp.hdr.nCur = nR + nC + nL + nV + nS + nD;

  // ---- 2. zero-free diagonal ------------------------------------------------------------------
  // 2a. STRUCTURED matching (numerically safe static pivots, replaces what partial pivoting does in solveReal.ts:15-34):
  // every source k is paired with one of its non-ground terminals n_k by a bipartite matching sources <-> nodes; column
  // n_k takes the branch row j_k (pivot +-1), column j_k takes the KCL row of n_k (pivot +-1), every other node keeps its
  // own KCL row.  A GROUNDED source's two pivots can never be touched by another elimination (its branch row and branch
  // column hold a single entry each), so it may sit anywhere in the order.  A FLOATING source's pair is eliminated before
  // everything else (all node columns n_k, then all branch columns j_k): the branch rows restricted to the matched nodes
  // are the incidence matrix of a forest with a unique perfect matching, so every leading minor is +-1 and both pivots of
  // every pair are exactly +-1 in any order; what remains is the conductance matrix of the circuit with those sources
  // contracted (v(n_k) = v(m_k) + V_k) — a symmetric M-matrix, safe under any diagonal pivot order.  (With a generic
  // transversal and the pair left inside the nested dissection, eliminating a neighbour first turned the +-1 pivot into
  // 1 - g/g = 0: `V1 a b / R1 a b / R2 a 0` was reported singular.)
  std::vector<int> row_of_col(n, -1);
  std::vector<int> forced;  // vertices eliminated first: matched nodes of floating sources, then their branch unknowns
  bool structured = true;
  {
    std::vector<int> node_of_src(nV, -1), src_of_node(nN, -1);
    std::vector<std::vector<int>> src_at(nN);
    std::vector<std::array<int, 2>> term(nV);
    std::vector<int> deg(nN, 0);
    for (int c = 0; c < nN; c++) deg[c] = (int)arow[c].size();
    for (int k = 0; k < nV; k++) {
      int i1 = d->V_n1[k] - 1, i2 = d->V_n2[k] - 1;
      if (i1 == i2) { structured = false; break; }  // shorted source: empty branch row
      // lower-degree terminal first: eliminating the matched node early makes a clique of its neighbours
      if (i1 >= 0 && i2 >= 0 && deg[i2] < deg[i1]) std::swap(i1, i2);
      if (i1 < 0) std::swap(i1, i2);
      term[k] = {i1, i2};
      if (i1 >= 0) src_at[i1].push_back(k);
      if (i2 >= 0) src_at[i2].push_back(k);
    }
    std::vector<int> seen_s(nV, -1), seen_n(nN, -1);
    // augment from a node: the node takes one of its sources, which may push that source's node elsewhere
    std::function<bool(int, int)> try_node = [&](int node, int stamp) -> bool {
      for (int s2 : src_at[node]) {
        if (seen_s[s2] == stamp) continue;
        seen_s[s2] = stamp;
        const int other = node_of_src[s2];
        if (other < 0 || try_node(other, stamp)) { node_of_src[s2] = node; src_of_node[node] = s2; return true; }
      }
      return false;
    };
    std::function<bool(int, int)> try_src = [&](int s2, int stamp) -> bool {
      for (int t : term[s2]) {
        if (t < 0 || seen_n[t] == stamp) continue;
        seen_n[t] = stamp;
        const int other = src_of_node[t];
        if (other < 0 || try_src(other, stamp)) { node_of_src[s2] = t; src_of_node[t] = s2; return true; }
      }
      return false;
    };
    if (structured) {
      // nodes without a structural diagonal (only sources attached) must be a source's matched node
      for (int c = 0; c < nN && structured; c++)
        if (!std::binary_search(arow[c].begin(), arow[c].end(), c) && !try_node(c, c)) structured = false;
      for (int k = 0; k < nV && structured; k++)
        if (node_of_src[k] < 0 && !try_src(k, k)) structured = false;
    }
    if (structured) {
      for (int c = 0; c < nN; c++) row_of_col[c] = src_of_node[c] >= 0 ? nN + src_of_node[c] : c;
      for (int k = 0; k < nV; k++) row_of_col[nN + k] = node_of_src[k];
      for (int k = 0; k < nV; k++)
        if (term[k][0] >= 0 && term[k][1] >= 0) forced.push_back(node_of_src[k]);
      const size_t nf = forced.size();
      for (size_t i = 0; i < nf; i++) forced.push_back(nN + src_of_node[forced[i]]);
    }
  }
  // 2b. anything the structured matching cannot express (it fails exactly when sources form a loop or a node hangs on
  // nothing but an over-subscribed source): generic maximum transversal; a failure there is a structurally singular matrix
  if (!structured && !max_transversal(n, cand, row_of_col)) {
    hp.structurally_singular = true;
    // keep a trivially valid (identity) program so that the handle can exist; run() reports singular
    row_of_col.resize(n);
    std::iota(row_of_col.begin(), row_of_col.end(), 0);
    for (int c = 0; c < n; c++) { arow[c].push_back(c); sort_unique(arow[c]); }
  }
  std::vector<int> col_of_row(n);
  for (int c = 0; c < n; c++) col_of_row[row_of_col[c]] = c;

  // ---- 3. ordering on the symmetrised pattern of B = P A (B[c][*] = A[row_of_col[c]][*]) -------
  Adj g(n);
  for (int c = 0; c < n; c++)
    for (int c2 : arow[row_of_col[c]])
      if (c2 != c) { g[c].push_back(c2); g[c2].push_back(c); }
  for (int c = 0; c < n; c++) sort_unique(g[c]);
  // the forced vertices leave the graph first (their fill joins their remaining neighbours); nested dissection orders the rest
  Adj g2 = g;
  std::vector<char> gone(n, 0);
  for (int v : forced) {
    std::vector<int> nb;
    for (int w : g2[v])
      if (!gone[w]) nb.push_back(w);
    for (int a : nb)
      for (int b : nb)
        if (a != b) g2[a].push_back(b);
    for (int a : nb) sort_unique(g2[a]);
    gone[v] = 1;
  }
  NDOrder nd(g2);
  for (int v : forced) nd.owner[v] = -1;
  nd.run();
  if (nd.order.size() + forced.size() != (size_t)n) { err = "internal: ordering lost vertices"; return SPICEY_ERR_BAD_DESC; }
  hp.cpos.assign(n, -1);
  {
    int p = 0;
    for (int v : forced) hp.cpos[v] = p++;
    for (int v : nd.order) hp.cpos[v] = p++;
  }
  hp.rpos.assign(n, -1);
  for (int r = 0; r < n; r++) hp.rpos[r] = hp.cpos[col_of_row[r]];

  // ---- 4. symbolic factorisation (symmetric pattern), etree, levels ----------------------------
  Adj upper(n);  // struct of row k right of the diagonal = struct of column k below it
  for (int c = 0; c < n; c++)
    for (int c2 : g[c]) {
      int a = hp.cpos[c], b = hp.cpos[c2];
      if (b > a) upper[a].push_back(b);
    }
  hp.parent.assign(n, -1);
  Adj children(n);
  for (int k = 0; k < n; k++) {
    std::vector<int> &S = upper[k];
    for (int c : children[k])
      for (int j : upper[c])
        if (j != k) S.push_back(j);
    sort_unique(S);
    if (!S.empty()) {
      hp.parent[k] = S[0];
      children[S[0]].push_back(k);
    }
  }
  hp.level.assign(n, 0);
  int nLevels = 1;
  for (int k = 0; k < n; k++) {
    int l = 0;
    for (int c : children[k]) l = std::max(l, hp.level[c] + 1);
    hp.level[k] = l;
    nLevels = std::max(nLevels, l + 1);
  }
  // CSR of L+U rows
  EntryIndex E;
  {
    Adj rows(n);
    for (int k = 0; k < n; k++) {
      rows[k].push_back(k);
      for (int j : upper[k]) { rows[k].push_back(j); rows[j].push_back(k); }
    }
    E.ptr.assign(n + 1, 0);
    for (int r = 0; r < n; r++) {
      sort_unique(rows[r]);
      E.ptr[r + 1] = E.ptr[r] + (int)rows[r].size();
    }
    E.col.reserve(E.ptr[n]);
    for (int r = 0; r < n; r++) E.col.insert(E.col.end(), rows[r].begin(), rows[r].end());
  }
  const int nLU = E.ptr[n];
  hp.hdr.nLU = nLU; hp.hdr.nW = nLU + n; hp.hdr.nLevels = nLevels;  // (nW grows by the constant-one slot when fronts are on, step 4b)
  {
    // classes of the CSR positions
    std::vector<uint8_t> is_dyn(nLU, 0), is_tgt(nLU, 0);
    auto mark2 = [&](const int32_t *a, const int32_t *b, int cnt) {
      for (int i = 0; i < cnt; i++) {
        const int i1 = a[i] - 1, i2 = b[i] - 1;
        auto m = [&](int r, int c) { const int p = E.pos(hp.rpos[r], hp.cpos[c]); if (p >= 0) is_dyn[p] = 1; };
        if (i1 >= 0) m(i1, i1);
        if (i2 >= 0) m(i2, i2);
        if (i1 >= 0 && i2 >= 0) { m(i1, i2); m(i2, i1); }
      }
    };
    mark2(d->S_n1, d->S_n2, nS);
    mark2(d->D_np, d->D_nm, nD);
    for (int k = 0; k < n; k++)
      for (int a : upper[k])
        for (int b : upper[k]) is_tgt[E.pos(a, b)] = 1;
    // Numbering inside each class: SLOT-MAJOR over the pivots of a level — (level of the owning pivot m = min(row,
    // col), diagonal / L / U, position of the other index in upper[m], m).  The factor tasks of a phase are ordered
    // by (slot pair, pivot), so the 32 lanes of a half-wave read the same slot of 32 consecutive pivots: consecutive
    // addresses, distinct LDS banks for the L, d and U operands (in CSR order they were 2.6-way conflicted on the
    // chain; only the scattered targets still are).  slot_major off = the old CSR order (spicey_build_program compiles both and keeps the cheaper one).
    E.id_of_pos.assign(nLU, -1);
    {
      std::vector<int> posr(nLU);
      for (int r = 0; r < n; r++)
        for (int p = E.ptr[r]; p < E.ptr[r + 1]; p++) posr[p] = r;
      std::vector<std::array<int64_t, 2>> key(nLU);
      for (int p = 0; p < nLU; p++) {
        const int r = posr[p], c = E.col[p], m = std::min(r, c), o = std::max(r, c);
        const int cls = is_dyn[p] ? 0 : (is_tgt[p] ? 1 : 2);
        int64_t slot = 0;
        if (o != m) slot = std::lower_bound(upper[m].begin(), upper[m].end(), o) - upper[m].begin();
        const int kind = r == c ? 0 : (c < r ? 1 : 2);
        if (slot_major) key[p] = {((int64_t)cls << 40) | ((int64_t)hp.level[m] << 20) | ((int64_t)kind << 18) | slot, (int64_t)m};
        else key[p] = {(int64_t)cls << 40, (int64_t)p};
      }
      std::vector<int> ordp(nLU);
      std::iota(ordp.begin(), ordp.end(), 0);
      std::stable_sort(ordp.begin(), ordp.end(), [&](int a, int b) { return key[a] < key[b]; });
      for (int i = 0; i < nLU; i++) E.id_of_pos[ordp[i]] = i;
    }
    int nrest = 0;
    for (int p = 0; p < nLU; p++) nrest += (is_dyn[p] || is_tgt[p]) ? 1 : 0;
    hp.hdr.nRestore = nrest;
    E.row_of_id.assign(nLU, 0); E.col_of_id.assign(nLU, 0);
    for (int r = 0; r < n; r++)
      for (int p = E.ptr[r]; p < E.ptr[r + 1]; p++) { E.row_of_id[E.id_of_pos[p]] = r; E.col_of_id[E.id_of_pos[p]] = E.col[p]; }
  }
  std::vector<int> diag(n);
  for (int k = 0; k < n; k++) diag[k] = E.find(k, k);

  // ---- 4b. dense fronts above the cut (multifrontal upper tree, fronts_exec.h) -------------------------------
  // Pivots of level >= Lc leave the level-scheduled task lists: consecutive pivots whose row structures nest
  // (upper[k-1] = {k} + upper[k]: a separator of the nested dissection) form one supernode = one dense front.
  int Lc = front_cut;
  if (Lc < 0) {
    // automatic: only where the long single-pivot chains of the top separators dominate (large, nonlinear circuits;
    // a linear circuit reuses its factors and keeps the task lists)
    // Up to 128 instances cannot fill the chip one workgroup each: there the fronts (with a group of workgroups per
    // instance) win from much smaller circuits on — measured on R/C/diode meshes, one instance: 20 x 20 (nnz(L+U) 7.5 k,
    // 51 levels) 0.151 -> 0.112 ms per step, 24 x 24 0.220 -> 0.134, 34 x 34 0.907 -> 0.168; 16 x 16 (4.3 k) is the tie.
    // Batches: 64 instances of the 34 x 34 mesh 0.893 -> 0.340 ms per step of the whole batch (4 workgroups each), 32:
    // 0.907 -> 0.271, 64 of the 24 x 24 mesh (which fits LDS) 0.278 -> 0.207; and a circuit too large for LDS gains even
    // with one workgroup per instance (256 x 34 x 34: 0.912 -> 0.814).
    // Where the circuit fits LDS, the 16-bit interpreter (one workgroup per instance) is the alternative and wins once a
    // group would be smaller than 8 workgroups (24 x 24: 32 instances 0.225 -> 0.186 with fronts, 64: 0.225 -> 0.209, 128:
    // 0.225 -> 0.304; 20 x 20 x 64: 0.152 -> 0.185): below ~16 k entries the fronts are for up to 32 instances only.
    const int64_t min_lu = d->n_inst <= 32 ? 6000 : 16000;
    // Level of the cut: 10, and 11 from ~7 000 unknowns on (round 3, after the front phases were rewritten for instruction
    // count; R/C/diode meshes, one instance, ms per step at cut 9 / 10 / 11 / 12: 34 x 34 0.156 / 0.146 / 0.150 / 0.153,
    // 70 x 70 0.241 / 0.232 / 0.238 / 0.241, 85 x 85 - / 0.323 / 0.321 / 0.332, 100 x 100 0.379 / 0.381 / 0.368 / 0.376,
    // 120 x 120 - / 0.551 / 0.550 / 0.557): a lower cut turns the last sparse levels into many one-panel fronts (6 - 9 us
    // each plus a hand-over on the chain that ends the sweep), a higher one adds workgroup-local levels below it.
    Lc = (nD + nS > 0 && nLU >= min_lu && nLevels > 24 && !hp.structurally_singular) ? (n >= 7000 ? 11 : 10) : 0;
    if (const char *e = getenv("SPICEY_FRONT_CUT")) Lc = atoi(e);  // experiments
  }
  if (Lc >= nLevels || hp.structurally_singular) Lc = 0;
  std::vector<int> front_of(n, -1);
  hp.fronts.clear(); hp.fr_asm.clear(); hp.fr_bnd.clear(); hp.fr_child.clear(); hp.fr_rel.clear(); hp.front_work.clear();
  if (Lc > 0) {
    auto pad16 = [](int x) { return (x + 15) & ~15; };
    const bool relax_fronts = !getenv("SPICEY_FRONT_EXACT");  // experiments: exact supernodes only
    int staged_mp = 176;
    if (const char *e = getenv("SPICEY_STAGED_MERGE_MP")) staged_mp = atoi(e);  // experiments (0: off)
    for (int k = 0; k < n; k++) {
      if (hp.level[k] < Lc) continue;
      // exact nesting (upper[k-1] = {k} + upper[k]) always merges; a RELAXED merge also takes a chain pivot whose row is
      // a little shorter than its parent's (the missing positions become explicit zeros of the dense front): every front
      // costs a fixed ~10 us of assembly / hand-over latency whatever its size, and the small chain fronts are the many.
      // Relaxed merges stop where the front would outgrow LDS residency (128 padded rows).
      bool merge = k > 0 && front_of[k - 1] >= 0 && !upper[k - 1].empty() && upper[k - 1][0] == k;
      if (merge) {
        const size_t grow = upper[k].size() + 1 - upper[k - 1].size();  // (upper[k-1] \ {k} is a subset of upper[k])
        const int pp = hp.fronts[front_of[k - 1]].p;
        const bool exact = grow == 0;
        const bool relaxed = relax_fronts && grow <= std::max<size_t>(4, upper[k].size() / 4) &&
                             pad16(pp + 1) + pad16((int)upper[k].size()) <= 128;
        // A front that is staged through the workspace anyway (beyond LDS residency on its own) still merges with its
        // chain parent while the result stays within `staged_mp` padded rows: one front's fixed cost less, and the
        // pivots of both pad to 16 once (13 + 33 pivots: 3 panels instead of 1 + 3)
        const bool staged = relax_fronts && staged_mp > 0 && grow <= std::max<size_t>(4, upper[k].size() / 4) &&
                            pad16(pp) + pad16((int)upper[k - 1].size()) > 128 &&
                            pad16(pp + 1) + pad16((int)upper[k].size()) <= staged_mp;
        merge = exact || relaxed || staged;
      }
      if (merge) {
        front_of[k] = front_of[k - 1];
        hp.fronts[front_of[k]].p++;
      } else {
        SpiceyFront f{};
        f.k0 = k; f.p = 1;
        front_of[k] = (int)hp.fronts.size();
        hp.fronts.push_back(f);
      }
    }
    bool ok = true;
    uint64_t off = 0;
    int max_mp = 0;
    std::vector<std::vector<int>> kids(hp.fronts.size());
    for (size_t fi = 0; fi < hp.fronts.size() && ok; fi++) {
      SpiceyFront &f = hp.fronts[fi];
      const std::vector<int> &B = upper[f.k0 + f.p - 1];
      f.q = (int)B.size();
      f.Pp = pad16(f.p); f.Mp = f.Pp + pad16(f.q); f.ld = f.Mp + 16;
      f.parent = f.q ? front_of[B[0]] : -1;
      if (f.q && f.parent < 0) ok = false;
      if (f.parent >= 0) kids[f.parent].push_back((int)fi);
      f.off = (uint32_t)off;
      off += (uint64_t)f.Mp * f.ld;
      max_mp = std::max(max_mp, f.Mp);
      f.bnd0 = (uint32_t)hp.fr_bnd.size();
      for (int b : B) hp.fr_bnd.push_back((uint32_t)b);
      auto loc = [&](int x) -> int {  // local index of pivot position x in this front
        if (x >= f.k0 && x < f.k0 + f.p) return x - f.k0;
        auto it = std::lower_bound(B.begin(), B.end(), x);
        if (it == B.end() || *it != x) { ok = false; return 0; }
        return f.Pp + (int)(it - B.begin());
      };
      f.asm0 = (uint32_t)(hp.fr_asm.size() / 2);
      double work = 0;
      for (int i = 0; i < f.p; i++) {
        const int k = f.k0 + i;
        hp.fr_asm.push_back((uint32_t)diag[k]); hp.fr_asm.push_back(((uint32_t)i << 16) | (uint32_t)i);  // row << 16 | column
        for (int b : upper[k]) {
          const int lb = loc(b);
          hp.fr_asm.push_back((uint32_t)E.find(k, b)); hp.fr_asm.push_back(((uint32_t)i << 16) | (uint32_t)lb);
          hp.fr_asm.push_back((uint32_t)E.find(b, k)); hp.fr_asm.push_back(((uint32_t)lb << 16) | (uint32_t)i);
        }
        hp.fr_asm.push_back((uint32_t)(nLU + k)); hp.fr_asm.push_back(((uint32_t)i << 16) | (uint32_t)f.Mp);
        work += (double)upper[k].size() * (double)(upper[k].size() + 1);
      }
      f.asm_n = (uint32_t)(hp.fr_asm.size() / 2) - f.asm0;
      // schedule weight in shader cycles: trailing updates at the MFMA rate (64 multiply-adds per cycle per CU), plus the
      // measured fixed costs — ~5 us per front (assembly, hand-over, store: latency of a few dependent L2 round trips),
      // ~3.5 us per panel (diagonal block, triangular solves), and the staged path of a front too large for LDS
      hp.front_work.push_back(work / 64.0 + 12000.0 + 8000.0 * (f.Pp / 16) + (f.Mp > 128 ? 60000.0 : 0.0));
    }
    if (off >= ((uint64_t)1 << 31) || max_mp > 448) ok = false;  // 32-bit offsets; panels of the largest front must fit LDS
    // A front adds its children's blocks in a fixed order and waits for each in turn: the child expected LAST (the longest
    // chain of weights below it — a property of the tree, not of the schedule) goes last, so that the others are in by
    // the time it arrives
    if (ok && !getenv("SPICEY_FRONT_CHILD_ORDER_ID")) {
      std::vector<double> fin(hp.fronts.size(), 0.0);
      for (size_t fi = 0; fi < hp.fronts.size(); fi++) {  // children precede parents
        double last = 0.0;
        for (int c : kids[fi]) last = std::max(last, fin[c]);
        fin[fi] = last + hp.front_work[fi];
        std::stable_sort(kids[fi].begin(), kids[fi].end(), [&](int a, int b) { return fin[a] < fin[b]; });
      }
    }
    for (size_t fi = 0; fi < hp.fronts.size() && ok; fi++) {
      SpiceyFront &f = hp.fronts[fi];
      f.child0 = (uint32_t)hp.fr_child.size();
      f.child_n = (uint32_t)kids[fi].size();
      for (int c : kids[fi]) hp.fr_child.push_back((uint32_t)c);
      f.rel0 = (uint32_t)hp.fr_rel.size();
      if (f.parent >= 0) {
        const SpiceyFront &pf = hp.fronts[f.parent];
        const std::vector<int> &PB = upper[pf.k0 + pf.p - 1];
        for (int j = 0; j < f.q; j++) {
          const int b = (int)hp.fr_bnd[f.bnd0 + j];
          int l;
          if (b >= pf.k0 && b < pf.k0 + pf.p) l = b - pf.k0;
          else {
            auto it = std::lower_bound(PB.begin(), PB.end(), b);
            if (it == PB.end() || *it != b) { ok = false; break; }
            l = pf.Pp + (int)(it - PB.begin());
          }
          hp.fr_rel.push_back((uint32_t)l);
        }
      }
    }
    if (!ok) {  // structure the dense kernels cannot take: keep the task lists for everything
      Lc = 0;
      hp.fronts.clear(); hp.fr_asm.clear(); hp.fr_bnd.clear(); hp.fr_child.clear(); hp.fr_rel.clear(); hp.front_work.clear();
      std::fill(front_of.begin(), front_of.end(), -1);
    } else {
      hp.hdr.front_ws = (int64_t)off;
      hp.hdr.max_front_mp = max_mp;
    }
  }
  hp.hdr.nFronts = (int32_t)hp.fronts.size();
  hp.hdr.front_cut = Lc;
  hp.hdr.one_slot = nLU + n;
  if (Lc > 0) hp.hdr.nW = nLU + n + 1;  // + the constant-one slot
  // Subtree-local levels below the cut (program.h): bin_of[k] for every pivot below the cut.  The subtrees (rooted where
  // the parent's level reaches the cut) are dealt into bins largest first, each to the lightest bin so far.
  std::vector<int> bin_of(n, -1);
  int nBins = 0;
  if (Lc > 0) {
    int want = 128;
    if (const char *e = getenv("SPICEY_BINS")) want = atoi(e);  // experiments (0: every level below the cut is a group phase)
    std::vector<int> sub_of(n, -1);
    std::vector<double> sub_work;
    for (int k = n - 1; k >= 0; k--) {  // parents are numbered behind their children
      if (hp.level[k] >= Lc) continue;
      const int p = hp.parent[k];
      if (p < 0 || hp.level[p] >= Lc) { sub_of[k] = (int)sub_work.size(); sub_work.push_back(0.0); }
      else sub_of[k] = sub_of[p];
      const double u = (double)upper[k].size();
      sub_work[sub_of[k]] += u * (u + 1.0) + 4.0;
    }
    nBins = std::min<int>(want, (int)sub_work.size());
    if (nBins > 0) {
      std::vector<int> order(sub_work.size());
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sub_work[a] > sub_work[b]; });
      std::vector<double> load(nBins, 0.0);
      std::vector<int> bin_of_sub(sub_work.size(), 0);
      for (int sidx : order) {
        const int b = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        bin_of_sub[sidx] = b;
        load[b] += sub_work[sidx];
      }
      for (int k = 0; k < n; k++) if (sub_of[k] >= 0) bin_of[k] = bin_of_sub[sub_of[k]];
    }
  }
  hp.hdr.nBins = nBins;
  hp.bin_upd.clear(); hp.bin_bk.clear();

  // ---- 5a. stamp lists --------------------------------------------------------------------------
  auto ent = [&](int r_orig, int c_orig) { return E.find(hp.rpos[r_orig], hp.cpos[c_orig]); };
  std::vector<std::vector<uint32_t>> stat(nLU), dyn(nLU);
  bool lookup_failed = false;
  auto stamp2 = [&](std::vector<std::vector<uint32_t>> &dst, int n1, int n2, uint32_t gi) {
    int i1 = n1 - 1, i2 = n2 - 1;
    auto put = [&](int r, int c, uint32_t v) {
      int e = ent(r, c);
      if (e < 0) { lookup_failed = true; return; }
      dst[e].push_back(v);
    };
    if (i1 >= 0) put(i1, i1, gi);
    if (i2 >= 0) put(i2, i2, gi);
    if (i1 >= 0 && i2 >= 0) { put(i1, i2, gi | SPICEY_NEG); put(i2, i1, gi | SPICEY_NEG); }
  };
  for (int i = 0; i < nR; i++) stamp2(stat, d->R_n1[i], d->R_n2[i], (uint32_t)i);
  for (int i = 0; i < nC; i++) stamp2(stat, d->C_n1[i], d->C_n2[i], (uint32_t)(nR + i));
  for (int i = 0; i < nL; i++) stamp2(stat, d->L_n1[i], d->L_n2[i], (uint32_t)(nR + nC + i));
  for (int i = 0; i < nS; i++) stamp2(dyn, d->S_n1[i], d->S_n2[i], (uint32_t)i);
  if (!hp.structurally_singular) {
    const uint32_t one = (uint32_t)(nR + nC + nL);
    for (int k = 0; k < nV; k++) {  // stampVoltageSourceReal.ts:12-31
      int i1 = d->V_n1[k] - 1, i2 = d->V_n2[k] - 1, j = nN + k;
      if (i1 == i2) continue;
      auto put = [&](int r, int c, uint32_t v) {
        int e = ent(r, c);
        if (e < 0) { lookup_failed = true; return; }
        stat[e].push_back(v);
      };
      if (i1 >= 0) put(i1, j, one);
      if (i2 >= 0) put(i2, j, one | SPICEY_NEG);
      if (i1 >= 0) put(j, i1, one);
      if (i2 >= 0) put(j, i2, one | SPICEY_NEG);
    }
  }
  for (int i = 0; i < nD; i++) stamp2(dyn, d->D_np[i], d->D_nm[i], (uint32_t)(nS + i));
  if (lookup_failed) { err = "internal: stamp outside the symbolic pattern"; return SPICEY_ERR_BAD_DESC; }

  hp.ent_flag.assign(nLU, 0);
  for (int k = 0; k < n; k++)
    if (hp.level[k] == 0) hp.ent_flag[diag[k]] |= 1;
  hp.stat_ptr.assign(1, 0);
  hp.stat_idx.clear();
  hp.dyn_ent.clear(); hp.dyn_ptr.assign(1, 0); hp.dyn_idx.clear();
  for (int e = 0; e < nLU; e++) {
    hp.stat_idx.insert(hp.stat_idx.end(), stat[e].begin(), stat[e].end());
    hp.stat_ptr.push_back((uint32_t)hp.stat_idx.size());
    if (!dyn[e].empty()) {
      hp.ent_flag[e] |= 2;
      hp.dyn_ent.push_back((uint32_t)e | ((hp.ent_flag[e] & 1) ? SPICEY_TGT_RECIP : 0u));
      hp.dyn_idx.insert(hp.dyn_idx.end(), dyn[e].begin(), dyn[e].end());
      hp.dyn_ptr.push_back((uint32_t)hp.dyn_idx.size());
    }
  }
  hp.hdr.nDynEnt = (int32_t)hp.dyn_ent.size();

  // right-hand side (stampCurrentReal.ts:3-14: b[i+] -= I, b[i-] += I; order C, L, V, D)
  {
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> rows(n);  // (u idx|sign, gstat coef idx)
    const uint32_t one = (uint32_t)(nR + nC + nL);
    auto cur = [&](int np, int nm, uint32_t ui, uint32_t cof, bool negate_value) {
      // contributes  -I to row np-1 and +I to row nm-1, where I = (negate_value ? -1 : 1) * cof * u[ui]
      int ip = np - 1, im = nm - 1;
      if (ip >= 0) rows[hp.rpos[ip]].emplace_back(ui | (negate_value ? 0u : SPICEY_NEG), cof);
      if (im >= 0) rows[hp.rpos[im]].emplace_back(ui | (negate_value ? SPICEY_NEG : 0u), cof);
    };
    for (int i = 0; i < nC; i++) cur(d->C_n1[i], d->C_n2[i], (uint32_t)i, (uint32_t)(nR + i), true);  // Ieq = -Gc*vPrev
    for (int i = 0; i < nL; i++) cur(d->L_n1[i], d->L_n2[i], (uint32_t)(nC + i), one, false);          // I = iPrev
    for (int k = 0; k < nV; k++) rows[hp.rpos[nN + k]].emplace_back((uint32_t)(nC + nL + k), one);     // b[j] += V
    for (int i = 0; i < nD; i++) cur(d->D_np[i], d->D_nm[i], (uint32_t)(nC + nL + nV + i), one, false);  // I = ieq
    hp.rhs_ptr.assign(1, 0);
    hp.rhs_idx.clear(); hp.rhs_cof.clear();
    for (int r = 0; r < n; r++) {
      for (auto &p : rows[r]) { hp.rhs_idx.push_back(p.first); hp.rhs_cof.push_back(p.second); }
      hp.rhs_ptr.push_back((uint32_t)hp.rhs_idx.size());
    }
    hp.hdr.nRhsIdx = (int32_t)hp.rhs_idx.size();
  }

  // ---- 5b. factorisation tasks per level ---------------------------------------------------------
  hp.lvl_slice.assign(1, 0);
  hp.upd_slice.clear(); hp.upd_tgt.clear(); hp.upd_cnt.clear(); hp.upd_pairs.clear();
  hp.n_products = 0;
  std::vector<std::vector<int>> by_level(nLevels);
  for (int k = 0; k < n; k++) by_level[hp.level[k]].push_back(k);
  std::map<uint32_t, std::vector<uint32_t>> iface;  // nBins > 0: products of the targets above the cut, by target
  // the pivot that owns a factor target: the smaller index of an entry, the row of a right-hand side
  auto owner_of = [&](uint32_t t) { return (int)t < nLU ? std::min(E.row_of_id[t], E.col_of_id[t]) : (int)t - nLU; };
  for (int l = 0; l < nLevels; l++) {
    // (target, pivot, L entry, U entry) tuples, grouped by target in pivot order
    struct Prod { uint32_t tgt, l, d, u; };
    std::vector<Prod> prods;
    if (Lc > 0 && l >= Lc) {  // factored as dense fronts
      if (l == Lc && nBins > 0) {
        // the targets above the cut: one task each with the products of every level below the cut, in level order
        std::vector<std::pair<uint32_t, std::vector<uint32_t>>> tasks;
        for (auto &kv : iface) { hp.n_products += (int64_t)kv.second.size() / 3; tasks.emplace_back(kv.first, std::move(kv.second)); }
        iface.clear();
        pack_slices<3>(tasks, hp.upd_slice, hp.upd_tgt, hp.upd_cnt, hp.upd_pairs);
      }
      hp.lvl_slice.push_back((uint32_t)hp.upd_slice.size());
      continue;
    }
    for (int k : by_level[l]) {
      const std::vector<int> &S = upper[k];
      for (int a : S) {
        uint32_t le = (uint32_t)E.find(a, k);
        for (int b : S) prods.push_back({(uint32_t)E.find(a, b), le, (uint32_t)diag[k], (uint32_t)E.find(k, b)});
        prods.push_back({(uint32_t)(nLU + a), le, (uint32_t)diag[k], (uint32_t)(nLU + k)});  // fused forward elimination
      }
    }
    std::stable_sort(prods.begin(), prods.end(), [](const Prod &x, const Prod &y) { return x.tgt < y.tgt; });
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> tasks;
    for (size_t i = 0; i < prods.size();) {
      size_t j = i;
      std::vector<uint32_t> flat;
      while (j < prods.size() && prods[j].tgt == prods[i].tgt) {
        flat.push_back(prods[j].l); flat.push_back(prods[j].d); flat.push_back(prods[j].u);
        j++;
      }
      uint32_t t = prods[i].tgt;
      if ((int)t < nLU) {
        // diagonal that becomes final now?
        const int r = E.row_of_id[t];
        if (E.col_of_id[t] == r && hp.level[r] == l + 1 && !(Lc > 0 && hp.level[r] >= Lc)) t |= SPICEY_TGT_RECIP;  // (a front inverts its own pivots)
      }
      tasks.emplace_back(t, std::move(flat));
      i = j;
    }
    if (nBins > 0) {
      std::vector<std::vector<std::pair<uint32_t, std::vector<uint32_t>>>> per_bin(nBins);
      for (auto &tk : tasks) {
        const int b = bin_of[owner_of(SPICEY_IDX(tk.first))];
        if (b < 0) {
          std::vector<uint32_t> &dst = iface[tk.first];
          dst.insert(dst.end(), tk.second.begin(), tk.second.end());
        } else {
          hp.n_products += (int64_t)tk.second.size() / 3;
          per_bin[b].push_back(std::move(tk));
        }
      }
      for (int b = 0; b < nBins; b++) {
        hp.bin_upd.push_back((uint32_t)hp.upd_slice.size());
        pack_slices<3>(per_bin[b], hp.upd_slice, hp.upd_tgt, hp.upd_cnt, hp.upd_pairs);
      }
      hp.bin_upd.push_back((uint32_t)hp.upd_slice.size());
      hp.lvl_slice.push_back((uint32_t)hp.upd_slice.size());
      continue;
    }
    hp.n_products += (int64_t)prods.size();
    pack_slices<3>(tasks, hp.upd_slice, hp.upd_tgt, hp.upd_cnt, hp.upd_pairs);
    hp.lvl_slice.push_back((uint32_t)hp.upd_slice.size());
  }
  hp.hdr.nUpdSlices = (int32_t)hp.upd_slice.size();

  // ---- 5c. backward substitution, v1 (32-bit) form: COLUMN-oriented -----------------------------------
  // x[k] = (y[k] - sum_b U[k][b] x[b]) / U[k][k] row by row means one thread walks a pivot's whole row; on a
  // mesh the top separator rows have > 100 entries and that serial walk dominated the step.  Instead, level by
  // level from the top, every row r below receives  y[r] -= U[r][k] * (y[k] * dinv[k])  from the pivots k of the
  // level (gather by target row: short lists, wide parallelism, the same 3-operand task as the factor levels);
  // one final phase scales x[i] = y[i] * dinv[i].  (The 16-bit records of the LDS path stay row-oriented: on
  // circuits that fit LDS the rows are short and one phase fewer matters more.)
  hp.bk_lvl_slice.assign(1, 0);
  hp.bk_slice.clear(); hp.bk_x.clear(); hp.bk_d.clear(); hp.bk_cnt.clear(); hp.bk_pairs.clear();
  hp.n_bk_products = 0;
  for (int l = 0; l < nLevels; l++) {
    struct Prod { uint32_t tgt, l, d, u; };
    std::vector<Prod> prods;
    if (Lc > 0 && l > Lc) { hp.bk_lvl_slice.push_back((uint32_t)hp.bk_slice.size()); continue; }
    // with fronts, "level Lc" holds the INTERFACE: every row below the cut receives its products with ALL upper
    // unknowns (solved by the fronts: W[nLU + k] = x[k], diagonal operand = the constant-one slot), highest level first
    const int l_hi = (Lc > 0 && l == Lc) ? nLevels - 1 : l;
    for (int ll = l_hi; ll >= l; ll--)
    for (int k : by_level[ll])
      for (int p = E.ptr[k]; p < E.ptr[k + 1]; p++) {
        const int r = E.col[p];
        if (r >= k) break;  // columns of row k left of the diagonal = rows r with U[r][k] != 0
        if (Lc > 0 && ll >= Lc) {
          if (hp.level[r] >= Lc) continue;  // upper rows are solved inside the fronts
          prods.push_back({(uint32_t)(nLU + r), (uint32_t)(nLU + k), (uint32_t)hp.hdr.one_slot, (uint32_t)E.find(r, k)});
          continue;
        }
        prods.push_back({(uint32_t)(nLU + r), (uint32_t)(nLU + k), (uint32_t)diag[k], (uint32_t)E.find(r, k)});
      }
    std::stable_sort(prods.begin(), prods.end(), [](const Prod &x, const Prod &y) { return x.tgt < y.tgt; });
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> tasks;
    for (size_t i = 0; i < prods.size();) {
      size_t j = i;
      std::vector<uint32_t> flat;
      while (j < prods.size() && prods[j].tgt == prods[i].tgt) { flat.push_back(prods[j].l); flat.push_back(prods[j].d); flat.push_back(prods[j].u); j++; }
      tasks.emplace_back(prods[i].tgt, std::move(flat));
      i = j;
    }
    hp.n_bk_products += (int64_t)prods.size();
    if (nBins > 0 && l <= Lc) {  // rows of one bin's subtrees together (program.h); l == Lc: the interface tasks, whose rows lie below the cut too
      std::vector<std::vector<std::pair<uint32_t, std::vector<uint32_t>>>> per_bin(nBins);
      for (auto &tk : tasks) per_bin[bin_of[(int)tk.first - nLU]].push_back(std::move(tk));
      for (int b = 0; b < nBins; b++) {
        hp.bin_bk.push_back((uint32_t)hp.bk_slice.size());
        pack_slices<3>(per_bin[b], hp.bk_slice, hp.bk_x, hp.bk_cnt, hp.bk_pairs);
      }
      hp.bin_bk.push_back((uint32_t)hp.bk_slice.size());
      hp.bk_lvl_slice.push_back((uint32_t)hp.bk_slice.size());
      continue;
    }
    pack_slices<3>(tasks, hp.bk_slice, hp.bk_x, hp.bk_cnt, hp.bk_pairs);
    hp.bk_lvl_slice.push_back((uint32_t)hp.bk_slice.size());
  }
  hp.hdr.nBkSlices = (int32_t)hp.bk_slice.size();
  hp.bk_d.assign(diag.begin(), diag.end());  // [n]: entry id of every pivot's (reciprocal) diagonal
  if (Lc > 0)
    for (int k = 0; k < n; k++)
      if (hp.level[k] >= Lc) hp.bk_d[k] = (uint32_t)hp.hdr.one_slot;  // the fronts leave x itself in W[nLU + k]

  // ---- 5c'. compact 16-bit records of the same tasks, phases in execution order -------------------
  hp.rec16.clear(); hp.ovf16.clear(); hp.ph_first.clear(); hp.ph_cnt.clear(); hp.ph_rhs.clear();
  hp.fus16.clear(); hp.fus_first.assign(nLevels, 0u); hp.fus_gen.assign(nLevels, 0u); hp.fus_rhs.assign(nLevels, 0u); hp.fus_pairs.assign(nLevels, 0u);
  hp.hdr.has16 = ((nLU + n) < 65535 && Lc == 0) ? 1 : 0;  // 0xFFFF = ground in the packed terminal words; fronts run under the 32-bit interpreter
  // Tridiagonal top: T = the pivots of the highest levels, at most 64 of them (one row per lane of a wave).  Two of them are coupled, once everything
  // below is eliminated, iff the original matrix couples them or some lower pivot has both in its row structure.  If that
  // coupling graph is a path, the Schur complement on T is tridiagonal in path order and the records stop below T.
  hp.pcr_tab.clear();
  hp.hdr.pcr_n = 0; hp.hdr.pcr_level = 0;
  std::vector<char> in_top(n, 0);
  int pcrL = 0;
  if (pcr_top && hp.hdr.has16 && nLevels >= 4 && !hp.structurally_singular) {
    int cnt = 0, L0 = nLevels;
    while (L0 > 1 && cnt + (int)by_level[L0 - 1].size() <= 64) { L0--; cnt += (int)by_level[L0].size(); }  // one row per lane of the solving wave
    if (cnt >= 15 && L0 >= 1 && L0 < nLevels) {
      std::vector<int> T;
      for (int k = 0; k < n; k++)
        if (hp.level[k] >= L0) { in_top[k] = 1; T.push_back(k); }
      std::vector<std::vector<int>> H(n);
      auto link = [&](int a, int b) { if (a != b) { H[a].push_back(b); H[b].push_back(a); } };
      for (int c = 0; c < n; c++)  // original couplings
        for (int c2 : g[c]) {
          const int a = hp.cpos[c], b = hp.cpos[c2];
          if (a < b && in_top[a] && in_top[b]) link(a, b);
        }
      for (int k = 0; k < n; k++) {  // fill through the pivots below
        if (in_top[k]) continue;
        std::vector<int> tk;
        for (int a : upper[k]) if (in_top[a]) tk.push_back(a);
        for (size_t i = 0; i < tk.size(); i++)
          for (size_t j = i + 1; j < tk.size(); j++) link(tk[i], tk[j]);
      }
      bool path = true;
      size_t edges = 0;
      int end0 = -1;
      for (int t : T) {
        sort_unique(H[t]);
        edges += H[t].size();
        if (H[t].size() > 2) path = false;
        if (H[t].size() <= 1 && (end0 < 0 || t < end0)) end0 = t;
      }
      if (edges != 2 * (T.size() - 1) || end0 < 0) path = false;
      std::vector<int> ord;
      if (path) {
        int prev = -1, cur = end0;
        while (cur >= 0) {
          ord.push_back(cur);
          int nxt = -1;
          for (int w : H[cur]) if (w != prev) nxt = w;
          prev = cur; cur = nxt;
          if (ord.size() > T.size()) { path = false; break; }
        }
        if (ord.size() != T.size()) path = false;  // (a cycle-free walk that covers T: connected)
      }
      if (path) {
        for (size_t i = 0; i < ord.size(); i++) {
          const int t = ord[i];
          const int a = i > 0 ? E.find(t, ord[i - 1]) : -2, c2 = i + 1 < ord.size() ? E.find(t, ord[i + 1]) : -2;
          if (a == -1 || c2 == -1) { path = false; break; }
          hp.pcr_tab.push_back(a < 0 ? (uint16_t)0xFFFF : (uint16_t)a);
          hp.pcr_tab.push_back((uint16_t)diag[t]);
          hp.pcr_tab.push_back(c2 < 0 ? (uint16_t)0xFFFF : (uint16_t)c2);
          hp.pcr_tab.push_back((uint16_t)(nLU + t));
        }
      }
      if (path) { hp.hdr.pcr_n = (int32_t)ord.size(); hp.hdr.pcr_level = L0; pcrL = L0; }
      else { hp.pcr_tab.clear(); std::fill(in_top.begin(), in_top.end(), 0); }
    }
  }
  if (pcrL == 0) std::fill(in_top.begin(), in_top.end(), 0);
  if (hp.hdr.has16) {
    auto emit_u = [&](uint32_t tgt, bool recip, const std::vector<uint32_t> &tr, std::vector<uint32_t> &dst) {  // tr = (l,d,u)*
      const uint32_t cnt = (uint32_t)(tr.size() / 3);
      uint32_t flags = SPICEY_R16_VALID | (recip ? SPICEY_R16_RECIP : 0u);
      uint32_t w0 = tgt | ((std::min(cnt, 255u) | (flags << 8)) << 16), w1 = 0, w2 = 0, w3 = 0;
      if (cnt <= 2) {
        if (cnt >= 1) { w1 = tr[0] | (tr[1] << 16); w2 = tr[2]; }
        if (cnt == 2) { w2 |= tr[3] << 16; w3 = tr[4] | (tr[5] << 16); }
      } else {
        w3 = (uint32_t)hp.ovf16.size();
        for (uint32_t v : tr) hp.ovf16.push_back((uint16_t)v);
      }
      dst.insert(dst.end(), {w0, w1, w2, w3});
    };
    auto emit_k = [&](uint32_t x, uint32_t dg, const std::vector<uint32_t> &pr) {  // pr = (u,xb)*
      const uint32_t cnt = (uint32_t)(pr.size() / 2);
      uint32_t flags = SPICEY_R16_VALID | SPICEY_R16_K;
      uint32_t w0 = x | ((std::min(cnt, 255u) | (flags << 8)) << 16), w1 = dg, w2 = 0, w3 = 0;
      if (cnt <= 2) {
        if (cnt >= 1) { w1 |= pr[0] << 16; w2 = pr[1]; }
        if (cnt == 2) { w2 |= pr[2] << 16; w3 = pr[3]; }
      } else {
        w3 = (uint32_t)hp.ovf16.size();
        for (uint32_t v : pr) hp.ovf16.push_back((uint16_t)v);
      }
      hp.rec16.insert(hp.rec16.end(), {w0, w1, w2, w3});
    };
    bool too_long = false;
    for (int l = 0; l < nLevels; l++) {  // factor phases: re-derive the grouped tasks of level l
      hp.ph_first.push_back((uint32_t)(hp.rec16.size() / 4));
      if (pcrL > 0 && l >= pcrL) { hp.ph_rhs.push_back(0u); hp.ph_cnt.push_back(0u); continue; }  // solved by cyclic reduction
      struct Prod { uint32_t tgt, l, d, u; int32_t si, sj, k; };  // si / sj: slots of the L / U operand in upper[k] (sj = -1: rhs)
      std::vector<Prod> prods;
      for (int k : by_level[l]) {
        const std::vector<int> &S = upper[k];
        for (int ia = 0; ia < (int)S.size(); ia++) {
          const int a = S[ia];
          uint32_t le = (uint32_t)E.find(a, k);
          for (int ib = 0; ib < (int)S.size(); ib++)
            prods.push_back({(uint32_t)E.find(a, S[ib]), le, (uint32_t)diag[k], (uint32_t)E.find(k, S[ib]), ia, ib, k});
          prods.push_back({(uint32_t)(nLU + a), le, (uint32_t)diag[k], (uint32_t)(nLU + k), ia, -1, k});
        }
      }
      // grouped by target in MATRIX-POSITION order (independent of the entry numbering, see spicey_build_program)
      auto poskey = [&](uint32_t t) -> uint64_t {
        return (int)t >= nLU ? (((uint64_t)1 << 62) | t) : (((uint64_t)E.row_of_id[t] << 31) | (uint64_t)E.col_of_id[t]);
      };
      std::stable_sort(prods.begin(), prods.end(), [&](const Prod &x, const Prod &y) { return poskey(x.tgt) < poskey(y.tgt); });
      struct UT { uint32_t t; bool recip; std::vector<uint32_t> tr; int32_t si, sj, k; };
      std::vector<UT> uts;
      for (size_t i = 0; i < prods.size();) {
        size_t j = i;
        std::vector<uint32_t> tr;
        while (j < prods.size() && prods[j].tgt == prods[i].tgt) { tr.push_back(prods[j].l); tr.push_back(prods[j].d); tr.push_back(prods[j].u); j++; }
        uint32_t t = prods[i].tgt;
        bool recip = false;
        if ((int)t < nLU) {
          const int r = E.row_of_id[t];
          recip = E.col_of_id[t] == r && hp.level[r] == l + 1 && !in_top[r];  // (the cyclic reduction wants the diagonal itself)
        }
        if (tr.size() / 3 > 255) too_long = true;
        uts.push_back({t, recip, std::move(tr), prods[i].si, prods[i].sj, prods[i].k});
        i = j;
      }
      // right-hand-side tasks first (a linear circuit's reused factorisation runs only those), then the same
      // (recip, count) next to each other: the 64-lane chunks then take one code path
      std::stable_sort(uts.begin(), uts.end(), [&](const UT &x, const UT &y) {
        const bool xr = (int)x.t >= nLU, yr = (int)y.t >= nLU;
        if (xr != yr) return xr;
        if (x.recip != y.recip) return x.recip > y.recip;
        if (x.tr.size() != y.tr.size()) return x.tr.size() > y.tr.size();
        if (!slot_major) return false;
        // lanes = the same operand slots of consecutive pivots (see the entry numbering): conflict-free L, d, U reads
        if (x.si != y.si) return x.si < y.si;
        if (x.sj != y.sj) return x.sj < y.sj;
        return x.k < y.k;
      });
      uint32_t nrhs = 0;
      for (auto &u : uts) nrhs += (int)u.t >= nLU ? 1u : 0u;
      hp.ph_rhs.push_back(nrhs);
      for (auto &u : uts) emit_u(u.t, u.recip, u.tr, hp.rec16);
      hp.ph_cnt.push_back((uint32_t)(hp.rec16.size() / 4) - hp.ph_first.back());
      // ---- the same phase as row records (program.h: fus16) -------------------------------------------------------------
      {
        // pivots of this level that reach row a; a row fits if it has <= 2 of them, each with <= 2 neighbours, and the
        // neighbours on the far side are distinct
        std::vector<std::vector<int>> reach(n);
        for (int k : by_level[l])
          for (int a : upper[k]) reach[a].push_back(k);
        auto other = [&](int k, int a) -> int {  // the neighbour of pivot k that is not a (-1: none)
          for (int b : upper[k]) if (b != a) return b;
          return -1;
        };
        std::vector<char> fits(n, 0);
        std::vector<int> rows;
        for (int a = 0; a < n; a++) {
          if (reach[a].empty() || reach[a].size() > 2) continue;
          bool ok = true;
          for (int k : reach[a]) ok = ok && upper[k].size() <= 2;
          if (ok && reach[a].size() == 2) {
            const int o0 = other(reach[a][0], a), o1 = other(reach[a][1], a);
            if (o0 >= 0 && o0 == o1) ok = false;
          }
          if (ok) { fits[a] = 1; rows.push_back(a); }
        }
        if (rows.size() >= 64) {  // (fewer: not worth a second encoding)
          hp.fus_first[l] = (uint32_t)(hp.fus16.size() / 4);
          uint32_t ngen = 0, ngrhs = 0;
          for (auto &u : uts) {
            const int row = (int)u.t >= nLU ? (int)u.t - nLU : E.row_of_id[u.t];
            if (fits[row]) continue;
            emit_u(u.t, u.recip, u.tr, hp.fus16);
            ngen++;
            ngrhs += (int)u.t >= nLU ? 1u : 0u;
          }
          hp.fus_gen[l] = ngen; hp.fus_rhs[l] = ngrhs;
          for (int a : rows) {
            std::sort(reach[a].begin(), reach[a].end());
            uint16_t h[16] = {0};
            const bool recip = hp.level[a] == l + 1 && !in_top[a];
            uint32_t meta = (uint32_t)reach[a].size() | ((SPICEY_R16_VALID | SPICEY_R16_FUSED | (recip ? SPICEY_R16_RECIP : 0u)) << 8);
            h[0] = (uint16_t)diag[a]; h[2] = (uint16_t)(nLU + a);
            for (size_t i = 0; i < reach[a].size(); i++) {
              const int k = reach[a][i], o = other(k, a);
              uint16_t *q = h + 3 + 6 * i;
              q[0] = (uint16_t)E.find(a, k); q[1] = (uint16_t)diag[k]; q[2] = (uint16_t)E.find(k, a); q[3] = (uint16_t)(nLU + k);
              if (o >= 0) { q[4] = (uint16_t)E.find(k, o); q[5] = (uint16_t)E.find(a, o); meta |= 1u << (4 + i); }
            }
            h[1] = (uint16_t)meta;
            for (int w = 0; w < 8; w++) hp.fus16.push_back((uint32_t)h[2 * w] | ((uint32_t)h[2 * w + 1] << 16));
          }
          hp.fus_pairs[l] = (uint32_t)rows.size();
        }
      }
    }
    for (int l = nLevels - 1; l >= 0; l--) {  // backward phases, top level first
      hp.ph_first.push_back((uint32_t)(hp.rec16.size() / 4));
      if (pcrL > 0 && l >= pcrL) { hp.ph_cnt.push_back(0u); continue; }
      std::vector<int> ks = by_level[l];
      std::stable_sort(ks.begin(), ks.end(), [&](int x, int y) { return upper[x].size() > upper[y].size(); });
      for (int k : ks) {
        std::vector<uint32_t> pr;
        for (int b : upper[k]) { pr.push_back((uint32_t)E.find(k, b)); pr.push_back((uint32_t)(nLU + b)); }
        if (pr.size() / 2 > 255) too_long = true;
        emit_k((uint32_t)(nLU + k), (uint32_t)diag[k], pr);
      }
      hp.ph_cnt.push_back((uint32_t)(hp.rec16.size() / 4) - hp.ph_first.back());
    }
    if (too_long || hp.ovf16.size() >= (size_t)1 << 31) {  // count field is 8 bits: such circuits use the 32-bit path
      hp.hdr.has16 = 0;
      hp.fus16.clear(); std::fill(hp.fus_pairs.begin(), hp.fus_pairs.end(), 0u);
      hp.hdr.pcr_n = 0; hp.hdr.pcr_level = 0; hp.pcr_tab.clear();
      hp.rec16.clear(); hp.ovf16.clear(); hp.ph_first.clear(); hp.ph_cnt.clear(); hp.ph_rhs.clear();
    }
  }
  hp.hdr.nRec16 = (int32_t)(hp.rec16.size() / 4);

  // ---- 5c''. v2 B-phase descriptors: per-entry dynamic stamps, per-row right-hand side -----------
  hp.ent_dd.assign(nLU, 0u);
  hp.dynx_ent.clear(); hp.dynx_ptr.assign(1, 0u); hp.dynx_idx.clear();
  for (int e = 0; e < nLU; e++) {
    uint32_t dd = (hp.ent_flag[e] & 1) && !dyn[e].empty() ? (1u << 30) : 0u;
    bool ovf = dyn[e].size() > 2;
    for (uint32_t v : dyn[e]) ovf = ovf || SPICEY_IDX(v) + 1 > 0x3fffu;
    if (ovf) {
      dd |= 1u << 31;
      hp.dynx_ent.push_back((uint32_t)e | ((hp.ent_flag[e] & 1) ? SPICEY_TGT_RECIP : 0u));
      hp.dynx_idx.insert(hp.dynx_idx.end(), dyn[e].begin(), dyn[e].end());
      hp.dynx_ptr.push_back((uint32_t)hp.dynx_idx.size());
    } else {
      for (size_t i = 0; i < dyn[e].size(); i++) {
        const uint32_t f = (SPICEY_IDX(dyn[e][i]) + 1) | ((dyn[e][i] & SPICEY_NEG) ? 0x4000u : 0u);
        dd |= f << (15 * i);
      }
    }
    hp.ent_dd[e] = dd;
  }
  hp.hdr.nDynX = (int32_t)hp.dynx_ent.size();
  hp.row_desc.assign((size_t)n * 2, 0u);
  hp.rowx.clear();
  for (int r = 0; r < n; r++) {
    const uint32_t j0 = hp.rhs_ptr[r], j1 = hp.rhs_ptr[r + 1];
    bool ovf = j1 - j0 > 4;
    for (uint32_t j = j0; j < j1; j++) ovf = ovf || SPICEY_IDX(hp.rhs_idx[j]) + 1 > 0x7fffu;
    if (ovf) {
      hp.row_desc[(size_t)r * 2 + 1] = 0xFFFFFFFFu;
      hp.rowx.push_back((uint32_t)r);
    } else {
      for (uint32_t j = j0; j < j1; j++) {
        const uint32_t f = (SPICEY_IDX(hp.rhs_idx[j]) + 1) | ((hp.rhs_idx[j] & SPICEY_NEG) ? 0x8000u : 0u);
        hp.row_desc[(size_t)r * 2 + (j - j0) / 2] |= f << (16 * ((j - j0) & 1));
      }
    }
  }
  hp.hdr.nRowX = (int32_t)hp.rowx.size();

  // ---- 5d. element terminals and outputs as W indices -------------------------------------------
  auto xpos = [&](int node) -> int32_t { return node == 0 ? -1 : (int32_t)(nLU + hp.cpos[node - 1]); };
  auto map2 = [&](const int32_t *a, const int32_t *b, int cnt, std::vector<int32_t> &oa, std::vector<int32_t> &ob) {
    oa.resize(cnt); ob.resize(cnt);
    for (int i = 0; i < cnt; i++) { oa[i] = xpos(a[i]); ob[i] = xpos(b[i]); }
  };
  map2(d->R_n1, d->R_n2, nR, hp.R_a, hp.R_b);
  map2(d->C_n1, d->C_n2, nC, hp.C_a, hp.C_b);
  map2(d->L_n1, d->L_n2, nL, hp.L_a, hp.L_b);
  map2(d->S_n1, d->S_n2, nS, hp.S_a, hp.S_b);
  map2(d->S_cp, d->S_cn, nS, hp.S_cp, hp.S_cn);
  map2(d->D_np, d->D_nm, nD, hp.D_a, hp.D_b);
  auto pack2 = [&](const std::vector<int32_t> &a, const std::vector<int32_t> &b, std::vector<uint32_t> &o) {
    o.resize(a.size());
    for (size_t i = 0; i < a.size(); i++)
      o[i] = (uint32_t)(a[i] < 0 ? 0xFFFF : (a[i] & 0xFFFF)) | ((uint32_t)(b[i] < 0 ? 0xFFFF : (b[i] & 0xFFFF)) << 16);
  };
  pack2(hp.R_a, hp.R_b, hp.R_ab); pack2(hp.C_a, hp.C_b, hp.C_ab); pack2(hp.L_a, hp.L_b, hp.L_ab); pack2(hp.D_a, hp.D_b, hp.D_ab);
  hp.V_x.resize(nV);
  for (int k = 0; k < nV; k++) hp.V_x[k] = nLU + hp.cpos[nN + k];
  hp.out_x.resize(nOut);
  for (int i = 0; i < nOut; i++) hp.out_x[i] = (d->n_out > 0 && d->out_nodes) ? xpos(d->out_nodes[i]) : (int32_t)(nLU + hp.cpos[i]);

  // natural numbering of entries and pivot positions (AC dense fallback)
  hp.pos_row.assign(n, 0); hp.pos_col.assign(n, 0);
  for (int r = 0; r < n; r++) hp.pos_row[hp.rpos[r]] = r;
  for (int c = 0; c < n; c++) hp.pos_col[hp.cpos[c]] = c;
  hp.ent_ro.assign(nLU, 0); hp.ent_co.assign(nLU, 0);
  for (int e = 0; e < nLU; e++) { hp.ent_ro[e] = hp.pos_row[E.row_of_id[e]]; hp.ent_co[e] = hp.pos_col[E.col_of_id[e]]; }
  // structural entries of A by natural column (diagnostics: program.h, col_ptr / col_ent)
  {
    std::vector<std::vector<uint32_t>> by_col(n);
    for (int e = 0; e < nLU; e++)
      if (!stat[e].empty() || !dyn[e].empty()) by_col[hp.ent_co[e]].push_back((uint32_t)e | ((hp.ent_flag[e] & 1) ? SPICEY_TGT_RECIP : 0u));
    hp.col_ptr.assign(1, 0u);
    hp.col_ent.clear();
    for (int c = 0; c < n; c++) {
      hp.col_ent.insert(hp.col_ent.end(), by_col[c].begin(), by_col[c].end());
      hp.col_ptr.push_back((uint32_t)hp.col_ent.size());
    }
  }

  // ---- 6. hybrid workspace layout (program.h: SpiceyProg::hybrid) ---------------------------------------------------------
  hp.hdr.hybrid = 0; hp.hdr.hyb_g0 = 0; hp.hdr.hyb_g2 = 0; hp.hdr.xoff = nLU;
  if (hybrid && slot_major && hp.hdr.has16 && Lc == 0 && nLevels >= 3 && !hp.structurally_singular) {
    // the leaf-owned entries: ids [0, g0) of the dynamic class and [nRestore, nRestore + g2) of the never-modified class
    // (inside a class the slot-major numbering sorts by the level of the owning pivot; a leaf-owned entry is never an
    // update target: targets lie among the ancestors of the eliminated pivot)
    const int nRest = hp.hdr.nRestore;
    auto leaf_owned = [&](int e) { return hp.level[std::min(E.row_of_id[e], E.col_of_id[e])] == 0; };
    int g0 = 0, g2 = 0;
    while (g0 < nRest && leaf_owned(g0)) g0++;
    while (nRest + g2 < nLU && leaf_owned(nRest + g2)) g2++;
    bool ok = true;
    for (int e = 0; e < nLU && ok; e++) ok = leaf_owned(e) == (e < g0 || (e >= nRest && e < nRest + g2));
    if (ok && g0 + g2 > 0) {
      const uint32_t G0 = (uint32_t)g0, G2 = (uint32_t)g2, NR = (uint32_t)nRest;
      auto is_glob = [&](uint32_t w) { return w < G0 || (w >= NR && w < NR + G2); };
      auto lds = [&](uint32_t w) -> uint32_t { return w < NR ? w - G0 : w - G0 - G2; };  // (w must not be a leaf-owned entry)
      bool bad = false;
      auto L16 = [&](uint32_t w) -> uint32_t { if (is_glob(w)) bad = true; return lds(w); };
      auto G16 = [&](uint32_t w) -> uint32_t { if (!is_glob(w)) bad = true; return w; };
      // 16-bit records of one phase, `count` of them from 16-byte unit `first` of `arr`; opg: the pivot operands are leaf-owned
      auto fix_generic = [&](std::vector<uint32_t> &arr, size_t first, size_t count, bool ktask, bool opg) {
        for (size_t i = 0; i < count; i++) {
          uint32_t *r = &arr[(first + i) * 4];
          const uint32_t meta = r[0] >> 16, cnt = meta & 0xffu;
          if (!(meta & (SPICEY_R16_VALID << 8))) continue;
          const uint32_t tgt_old = r[0] & 0xffffu;
          const bool rhs_task = tgt_old >= (uint32_t)nLU;
          r[0] = L16(tgt_old) | (meta << 16);
          if (ktask) {
            const uint32_t dg = r[1] & 0xffffu;
            const uint32_t dnew = opg ? G16(dg) : L16(dg);
            if (cnt <= 2) {
              uint32_t u0 = r[1] >> 16, x0 = r[2] & 0xffffu, u1 = r[2] >> 16, x1 = r[3] & 0xffffu;
              if (cnt >= 1) { u0 = opg ? G16(u0) : L16(u0); x0 = L16(x0); }
              if (cnt == 2) { u1 = opg ? G16(u1) : L16(u1); x1 = L16(x1); }
              r[1] = dnew | (u0 << 16); r[2] = x0 | (u1 << 16); r[3] = x1;
            } else {
              r[1] = dnew;
              for (uint32_t j = 0; j < cnt; j++) {
                uint16_t &u = hp.ovf16[r[3] + 2 * j], &x = hp.ovf16[r[3] + 2 * j + 1];
                u = (uint16_t)(opg ? G16(u) : L16(u)); x = (uint16_t)L16(x);
              }
            }
          } else {
            // third operand of a right-hand-side task: y_k, an LDS index in every phase
            auto third = [&](uint32_t u) { return (opg && !rhs_task) ? G16(u) : L16(u); };
            if (cnt <= 2) {
              uint32_t l0 = r[1] & 0xffffu, d0 = r[1] >> 16, u0 = r[2] & 0xffffu, l1 = r[2] >> 16, d1 = r[3] & 0xffffu, u1 = r[3] >> 16;
              if (cnt >= 1) { l0 = opg ? G16(l0) : L16(l0); d0 = opg ? G16(d0) : L16(d0); u0 = third(u0); }
              if (cnt == 2) { l1 = opg ? G16(l1) : L16(l1); d1 = opg ? G16(d1) : L16(d1); u1 = third(u1); }
              r[1] = l0 | (d0 << 16); r[2] = u0 | (l1 << 16); r[3] = d1 | (u1 << 16);
            } else {
              for (uint32_t j = 0; j < cnt; j++) {
                uint16_t *t = &hp.ovf16[r[3] + 3 * j];
                t[0] = (uint16_t)(opg ? G16(t[0]) : L16(t[0])); t[1] = (uint16_t)(opg ? G16(t[1]) : L16(t[1])); t[2] = (uint16_t)third(t[2]);
              }
            }
          }
        }
      };
      // (an overflow list belongs to ONE record of rec16 and, when the phase has a row-record encoding, to its twin in fus16,
      // which was emitted with its own copy: emit_u appends to ovf16 per call — so every list is fixed exactly once)
      for (int p = 0; p < 2 * nLevels; p++) {
        const bool ktask = p >= nLevels;
        const int lvl = ktask ? 2 * nLevels - 1 - p : p;
        fix_generic(hp.rec16, hp.ph_first[p], hp.ph_cnt[p], ktask, lvl == 0);
        if (!ktask && hp.fus_pairs[p] > 0) {
          fix_generic(hp.fus16, hp.fus_first[p], hp.fus_gen[p], false, lvl == 0);
          for (uint32_t i = 0; i < hp.fus_pairs[p]; i++) {
            uint32_t *w = &hp.fus16[((size_t)hp.fus_first[p] + hp.fus_gen[p]) * 4 + (size_t)i * 8];
            uint16_t h[16];
            for (int q = 0; q < 8; q++) { h[2 * q] = (uint16_t)(w[q] & 0xffffu); h[2 * q + 1] = (uint16_t)(w[q] >> 16); }
            const uint32_t meta = h[1], np = meta & 3u;
            h[0] = (uint16_t)L16(h[0]); h[2] = (uint16_t)L16(h[2]);
            for (uint32_t i2 = 0; i2 < np; i2++) {
              uint16_t *q = h + 3 + 6 * i2;
              const bool opg = lvl == 0;
              q[0] = (uint16_t)(opg ? G16(q[0]) : L16(q[0])); q[1] = (uint16_t)(opg ? G16(q[1]) : L16(q[1])); q[2] = (uint16_t)(opg ? G16(q[2]) : L16(q[2]));
              q[3] = (uint16_t)L16(q[3]);
              if ((meta >> (4 + i2)) & 1u) { q[4] = (uint16_t)(opg ? G16(q[4]) : L16(q[4])); q[5] = (uint16_t)L16(q[5]); }
            }
            for (int q = 0; q < 8; q++) w[q] = (uint32_t)h[2 * q] | ((uint32_t)h[2 * q + 1] << 16);
          }
        }
      }
      for (auto &v : hp.pcr_tab) if (v != 0xFFFFu) v = (uint16_t)L16(v);
      auto fix_i32 = [&](std::vector<int32_t> &a) { for (auto &v : a) if (v >= 0) v = (int32_t)L16((uint32_t)v); };
      fix_i32(hp.R_a); fix_i32(hp.R_b); fix_i32(hp.C_a); fix_i32(hp.C_b); fix_i32(hp.L_a); fix_i32(hp.L_b); fix_i32(hp.S_a); fix_i32(hp.S_b);
      fix_i32(hp.S_cp); fix_i32(hp.S_cn); fix_i32(hp.D_a); fix_i32(hp.D_b); fix_i32(hp.V_x); fix_i32(hp.out_x);
      auto fix_ab = [&](std::vector<uint32_t> &a) {
        for (auto &v : a) {
          uint32_t lo = v & 0xffffu, hi = v >> 16;
          if (lo != 0xFFFFu) lo = L16(lo);
          if (hi != 0xFFFFu) hi = L16(hi);
          v = lo | (hi << 16);
        }
      };
      fix_ab(hp.R_ab); fix_ab(hp.C_ab); fix_ab(hp.L_ab); fix_ab(hp.D_ab);
      if (bad) { err = "internal: hybrid layout met an index on the wrong side"; return SPICEY_ERR_BAD_DESC; }
      hp.hdr.hybrid = 1; hp.hdr.hyb_g0 = g0; hp.hdr.hyb_g2 = g2; hp.hdr.xoff = nLU - g0 - g2;
    }
  }

  hp.pack();
  return SPICEY_OK;
}

void HostProgram::pack() {
  blob.clear();
  offsets.clear();
  add_section(blob, offsets, stat_ptr);   // 0
  add_section(blob, offsets, stat_idx);   // 1
  add_section(blob, offsets, ent_flag);   // 2
  add_section(blob, offsets, dyn_ent);    // 3
  add_section(blob, offsets, dyn_ptr);    // 4
  add_section(blob, offsets, dyn_idx);    // 5
  add_section(blob, offsets, rhs_ptr);    // 6
  add_section(blob, offsets, rhs_idx);    // 7
  add_section(blob, offsets, rhs_cof);    // 8
  add_section(blob, offsets, lvl_slice);  // 9
  add_section(blob, offsets, upd_slice);  // 10
  add_section(blob, offsets, upd_tgt);    // 11
  add_section(blob, offsets, upd_cnt);    // 12
  add_section(blob, offsets, upd_pairs);  // 13
  add_section(blob, offsets, bk_lvl_slice);  // 14
  add_section(blob, offsets, bk_slice);   // 15
  add_section(blob, offsets, bk_x);       // 16
  add_section(blob, offsets, bk_d);       // 17
  add_section(blob, offsets, bk_cnt);     // 18
  add_section(blob, offsets, bk_pairs);   // 19
  add_section(blob, offsets, R_a); add_section(blob, offsets, R_b);    // 20 21
  add_section(blob, offsets, C_a); add_section(blob, offsets, C_b);    // 22 23
  add_section(blob, offsets, L_a); add_section(blob, offsets, L_b);    // 24 25
  add_section(blob, offsets, S_a); add_section(blob, offsets, S_b);    // 26 27
  add_section(blob, offsets, S_cp); add_section(blob, offsets, S_cn);  // 28 29
  add_section(blob, offsets, D_a); add_section(blob, offsets, D_b);    // 30 31
  add_section(blob, offsets, V_x);    // 32
  add_section(blob, offsets, out_x);  // 33
  add_section(blob, offsets, rec16);     // 34
  add_section(blob, offsets, ovf16);     // 35
  add_section(blob, offsets, ph_first);  // 36
  add_section(blob, offsets, ph_cnt);    // 37
  add_section(blob, offsets, ent_dd);    // 38
  add_section(blob, offsets, dynx_ent);  // 39
  add_section(blob, offsets, dynx_ptr);  // 40
  add_section(blob, offsets, dynx_idx);  // 41
  add_section(blob, offsets, row_desc);  // 42
  add_section(blob, offsets, rowx);      // 43
  add_section(blob, offsets, R_ab); add_section(blob, offsets, C_ab);  // 44 45
  add_section(blob, offsets, L_ab); add_section(blob, offsets, D_ab);  // 46 47
  add_section(blob, offsets, fronts);    // 48
  add_section(blob, offsets, fr_asm);    // 49
  add_section(blob, offsets, fr_bnd);    // 50
  add_section(blob, offsets, fr_child);  // 51
  add_section(blob, offsets, fr_rel);    // 52
  add_section(blob, offsets, pcr_tab);   // 53
  add_section(blob, offsets, fus16);     // 54
  add_section(blob, offsets, fus_first); add_section(blob, offsets, fus_gen);    // 55 56
  add_section(blob, offsets, fus_rhs); add_section(blob, offsets, fus_pairs);    // 57 58
  add_section(blob, offsets, ent_ro); add_section(blob, offsets, ent_co);        // 59 60
  add_section(blob, offsets, pos_row); add_section(blob, offsets, pos_col);      // 61 62
  add_section(blob, offsets, bin_upd); add_section(blob, offsets, bin_bk);       // 63 64
  add_section(blob, offsets, col_ptr); add_section(blob, offsets, col_ent);      // 65 66
  if (getenv("SPICEY_DUMP_SECTIONS")) {  // experiments: bytes per section of the program blob
    for (size_t i = 0; i < offsets.size(); i++) {
      const size_t end = i + 1 < offsets.size() ? offsets[i + 1] : blob.size();
      if (end - offsets[i] >= (1u << 16)) fprintf(stderr, "section %2zu: %9zu bytes\n", i, end - offsets[i]);
    }
  }
}

SpiceyProg HostProgram::bind(const void *base) const {
  SpiceyProg p = hdr;
  const uint8_t *b = (const uint8_t *)base;
  auto u32 = [&](int i) { return (const uint32_t *)(b + offsets[i]); };
  auto i32 = [&](int i) { return (const int32_t *)(b + offsets[i]); };
  p.stat_ptr = u32(0); p.stat_idx = u32(1); p.ent_flag = (const uint8_t *)(b + offsets[2]);
  p.dyn_ent = u32(3); p.dyn_ptr = u32(4); p.dyn_idx = u32(5);
  p.rhs_ptr = u32(6); p.rhs_idx = u32(7); p.rhs_cof = u32(8);
  p.lvl_slice = u32(9); p.upd_slice = (const SpiceySlice *)(b + offsets[10]);
  p.upd_tgt = u32(11); p.upd_cnt = u32(12); p.upd_pairs = u32(13);
  p.bk_lvl_slice = u32(14); p.bk_slice = (const SpiceySlice *)(b + offsets[15]);
  p.bk_x = u32(16); p.bk_d = u32(17); p.bk_cnt = u32(18); p.bk_pairs = u32(19);
  p.R_a = i32(20); p.R_b = i32(21); p.C_a = i32(22); p.C_b = i32(23); p.L_a = i32(24); p.L_b = i32(25);
  p.S_a = i32(26); p.S_b = i32(27); p.S_cp = i32(28); p.S_cn = i32(29); p.D_a = i32(30); p.D_b = i32(31);
  p.V_x = i32(32); p.out_x = i32(33);
  p.rec16 = u32(34); p.ovf16 = (const uint16_t *)(b + offsets[35]); p.ph_first = u32(36); p.ph_cnt = u32(37);
  p.ent_dd = u32(38); p.dynx_ent = u32(39); p.dynx_ptr = u32(40); p.dynx_idx = u32(41); p.row_desc = u32(42); p.rowx = u32(43);
  p.R_ab = u32(44); p.C_ab = u32(45); p.L_ab = u32(46); p.D_ab = u32(47);
  p.fr = (const SpiceyFront *)(b + offsets[48]);
  p.fr_asm = u32(49); p.fr_bnd = u32(50); p.fr_child = u32(51); p.fr_rel = u32(52);
  p.pcr_tab = (const uint16_t *)(b + offsets[53]);
  p.fus16 = u32(54); p.fus_first = u32(55); p.fus_gen = u32(56); p.fus_rhs = u32(57); p.fus_pairs = u32(58);
  p.ent_ro = i32(59); p.ent_co = i32(60); p.pos_row = i32(61); p.pos_col = i32(62);
  p.bin_upd = u32(63); p.bin_bk = u32(64);
  p.col_ptr = u32(65); p.col_ent = u32(66);
  return p;
}

// ---------------------------------------------------------------------------------------------
// Resident layout: chunks of 64 lanes; chunk c lives in (wave c % nWaves, slot c / nWaves), so the
// chunks of one phase spread over the waves.  Smallest phases first (they are pure latency).
void spicey_build_resident(const HostProgram &hp, int T, int rmax, HostResident &out, int max_tail, bool row_records) {
  out = HostResident();
  out.rmax = rmax; out.T = T;
  const int nPh = (int)hp.ph_cnt.size();
  const int nWaves = T / 64;
  out.res.assign((size_t)rmax * T * 4, 0u);
  out.res_phase.assign((size_t)nWaves * rmax, -1);
  out.st_first.assign(std::max(nPh, 1), 0u);
  out.st_cnt.assign(std::max(nPh, 1), 0u);
  out.st_rhs.assign(std::max(nPh, 1), 0u);
  out.st_fus.assign(std::max(nPh, 1), 0u);
  if (hp.hdr.has16 && nPh <= 254 && max_tail > 1 && hp.hdr.pcr_n == 0) {  // (a tridiagonal top replaces the tail: its phases hold no records)
    // tail: longest run of <= 64-task phases around the factor -> backward turn (phase nLevels-1 | nLevels)
    const int nL = hp.hdr.nLevels;
    int a = nL, b = nL;  // [a, b)
    // (hybrid workspace: the two phases of the leaves read their operands from the global array — they stay ordinary phases)
    const int a_min = hp.hdr.hybrid ? 1 : 0, b_max = hp.hdr.hybrid ? nPh - 1 : nPh;
    while (a > a_min && hp.ph_cnt[a - 1] <= 64 && b - (a - 1) <= max_tail) a--;
    while (b < b_max && hp.ph_cnt[b] <= 64 && (b + 1) - a <= max_tail) b++;
    while (a < b && hp.ph_cnt[a] == 0) a++;  // skip empty leading phases (the top factor level has no tasks)
    if (b - a >= 3) { out.tail_first = a; out.tail_n = b - a; }
  }
  if (hp.hdr.has16 && nPh <= 254) {
    // 1. which phases become resident: smallest first (they are pure latency) while chunks remain;
    //    chunk c of the running count goes to wave c % nWaves, so a phase spreads over the waves.
    //    A factor phase with a row-record encoding (program.h: fus16) is taken in that form: its generic remainder in
    //    one-slot chunks, its row records in chunks that own TWO consecutive slots of a wave (head + continuation).
    // (only where the generic form would give a thread more than one task: a level narrower than the workgroup is bound by
    // one LDS round trip either way, and a row record issues twice the operand reads of a task)
    auto rows_of = [&](int p) -> int { return (row_records && p < (int)hp.fus_pairs.size() && (int)hp.ph_cnt[p] > T) ? (int)hp.fus_pairs[p] : 0; };
    auto units_of = [&](int p) -> int { return rows_of(p) > 0 ? (int)hp.fus_gen[p] + 2 * rows_of(p) : (int)hp.ph_cnt[p]; };
    std::vector<int> order(nPh);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return units_of(a) < units_of(b); });
    struct Chunk { int phase, first, count, kind; };  // kind 0: generic records of rec16; 1: generic remainder of fus16; 2: row head; 3: row continuation
    std::vector<std::vector<Chunk>> per_wave(nWaves);
    std::vector<int> load(nWaves, 0);
    int next_chunk = 0;
    // tridiagonal top: the first backward phase below it goes to wave 0 — the wave that solves the top runs it right behind
    // the last stage (SpiceyResident::k_merge): one barrier phase less on the serial chain of every solve
    if (hp.hdr.pcr_n > 0 && rmax > 0 && !getenv("SPICEY_NO_KMERGE")) {
      const int pm = 2 * hp.hdr.nLevels - hp.hdr.pcr_level;
      if (pm > 0 && pm < nPh - 1 && hp.ph_cnt[pm] > 0 && hp.ph_cnt[pm] <= 64 && rows_of(pm) == 0) {
        per_wave[0].push_back({pm, 0, (int)hp.ph_cnt[pm], 0});
        load[0]++;
        out.k_merge = pm;
        out.resident_tasks += hp.ph_cnt[pm];
      }
    }
    for (int p : order) {
      const int cnt = (int)hp.ph_cnt[p];
      if (cnt == 0) continue;
      if (out.k_merge > 0 && p == out.k_merge) continue;  // (placed above)
      if (p >= out.tail_first && p < out.tail_first + out.tail_n) continue;  // lives in the LDS tail table
      const int nrow = rows_of(p), ngen = nrow > 0 ? (int)hp.fus_gen[p] : cnt;
      const int cg = (ngen + 63) / 64, cr = (nrow + 63) / 64;
      bool fits = true;
      {
        std::vector<int> l2 = load;
        for (int i = 0; i < cg; i++) if (++l2[(next_chunk + i) % nWaves] > rmax) fits = false;
        for (int i = 0; i < cr; i++) if ((l2[(next_chunk + cg + i) % nWaves] += 2) > rmax) fits = false;
      }
      if (!fits) {  // stays streamed
        out.st_first[p] = hp.ph_first[p];
        out.st_cnt[p] = hp.ph_cnt[p];
        out.st_rhs[p] = p < (int)hp.ph_rhs.size() ? hp.ph_rhs[p] : hp.ph_cnt[p];
        // streamed anyway: take the row records — where they keep most threads busy (a level with fewer rows than half the
        // workgroup is latency-bound on the one exposed record fetch; its generic records, several per thread, overlap theirs)
        out.st_fus[p] = (nrow > 0 && 2 * nrow > T) ? 1u : 0u;
        out.streamed_tasks += cnt;
        continue;
      }
      for (int i = 0; i < cg; i++) {
        const int w = (next_chunk + i) % nWaves;
        per_wave[w].push_back({p, i * 64, std::min(64, ngen - i * 64), nrow > 0 ? 1 : 0});
        load[w]++;
      }
      for (int i = 0; i < cr; i++) {
        const int w = (next_chunk + cg + i) % nWaves;
        per_wave[w].push_back({p, i * 64, std::min(64, nrow - i * 64), 2});
        per_wave[w].push_back({p, i * 64, std::min(64, nrow - i * 64), 3});
        load[w] += 2;
      }
      next_chunk += cg + cr;
      out.resident_tasks += cnt;
    }
    // 2. per wave: slots in execution (phase) order, so that the kernel walks them with a cursor (stable: a row chunk's
    //    continuation stays right behind its head)
    for (int w = 0; w < nWaves; w++) {
      std::stable_sort(per_wave[w].begin(), per_wave[w].end(), [](const Chunk &a, const Chunk &b) { return a.phase < b.phase; });
      for (size_t slot = 0; slot < per_wave[w].size(); slot++) {
        const Chunk &ch = per_wave[w][slot];
        out.res_phase[(size_t)w * rmax + slot] = ch.kind == 3 ? 0xFE : ch.phase;  // (0xFE: never a phase, nPh <= 254)
        for (int lane = 0; lane < ch.count; lane++) {
          const size_t dst = ((size_t)slot * T + (size_t)w * 64 + lane) * 4;
          size_t src;
          const std::vector<uint32_t> *from = &hp.fus16;
          if (ch.kind == 0) { from = &hp.rec16; src = ((size_t)hp.ph_first[ch.phase] + ch.first + lane) * 4; }
          else if (ch.kind == 1) src = ((size_t)hp.fus_first[ch.phase] + ch.first + lane) * 4;
          else src = ((size_t)hp.fus_first[ch.phase] + hp.fus_gen[ch.phase]) * 4 + (size_t)(ch.first + lane) * 8 + (ch.kind == 3 ? 4 : 0);
          for (int k = 0; k < 4; k++) out.res[dst + k] = (*from)[src + k];
        }
      }
    }
  } else if (hp.hdr.has16) {
    for (int p = 0; p < nPh; p++) {
      out.st_first[p] = hp.ph_first[p]; out.st_cnt[p] = hp.ph_cnt[p]; out.streamed_tasks += hp.ph_cnt[p];
      out.st_rhs[p] = p < (int)hp.ph_rhs.size() ? hp.ph_rhs[p] : hp.ph_cnt[p];
    }
  }
  // one descriptor per phase for the streamed path (program.h)
  out.st_desc.assign((size_t)std::max(nPh, 1) * 8, 0u);
  for (int p = 0; p < nPh; p++) {
    uint32_t *dsc = &out.st_desc[(size_t)p * 8];
    if (out.st_cnt[p] == 0) continue;
    if (out.st_fus[p]) {
      dsc[0] = 1u; dsc[1] = hp.fus_first[p] + hp.fus_gen[p]; dsc[2] = hp.fus_pairs[p]; dsc[3] = hp.fus_pairs[p];
      dsc[4] = hp.fus_first[p]; dsc[5] = hp.fus_gen[p]; dsc[6] = hp.fus_rhs[p];
    } else {
      dsc[0] = 0u; dsc[1] = out.st_first[p]; dsc[2] = out.st_cnt[p]; dsc[3] = out.st_rhs[p];
    }
  }
  out.pack();
}

void HostResident::pack() {
  blob.clear();
  offsets.clear();
  add_section(blob, offsets, res);
  add_section(blob, offsets, res_phase);
  add_section(blob, offsets, st_first);
  add_section(blob, offsets, st_cnt);
  add_section(blob, offsets, st_rhs);
  add_section(blob, offsets, st_fus);
  add_section(blob, offsets, st_desc);
}

SpiceyResident HostResident::bind(const void *base) const {
  SpiceyResident r{};
  const uint8_t *b = (const uint8_t *)base;
  r.res = (const uint32_t *)(b + offsets[0]);
  r.res_phase = (const int32_t *)(b + offsets[1]);
  r.st_first = (const uint32_t *)(b + offsets[2]);
  r.st_cnt = (const uint32_t *)(b + offsets[3]);
  r.st_rhs = (const uint32_t *)(b + offsets[4]);
  r.st_fus = (const uint32_t *)(b + offsets[5]);
  r.st_desc = (const uint32_t *)(b + offsets[6]);
  r.rmax = rmax;
  r.T = T;
  r.tail_first = tail_first;
  r.tail_n = tail_n;
  r.k_merge = k_merge;
  return r;
}

// ---------------------------------------------------------------------------------------------
// Front schedule: proportional mapping.  The roots share the G workgroups by subtree work; a front whose subtree owns
// the workgroup range [a, b) runs on workgroup a after its children, which split [a, b) among themselves by work
// (largest first; a child may get an empty share, it then runs on a too).  Lists are in global postorder, so a
// workgroup never waits for a front that is scheduled behind one of its own.
void spicey_build_front_schedule(const HostProgram &hp, int G, std::vector<uint32_t> &first, std::vector<uint32_t> &list) {
  const int nf = (int)hp.fronts.size();
  G = std::max(G, 1);
  first.assign((size_t)G + 1, 0u);
  list.clear();
  if (nf == 0) return;
  std::vector<double> sub(nf, 0.0);
  for (int f = 0; f < nf; f++) sub[f] = hp.front_work[f];
  for (int f = 0; f < nf; f++)  // children precede parents (fronts are numbered in pivot order)
    if (hp.fronts[f].parent >= 0) sub[hp.fronts[f].parent] += sub[f];
  std::vector<int> owner(nf, 0);
  std::vector<int> order;  // global postorder
  order.reserve(nf);
  struct Item { int f, a, b, stage; };
  std::vector<Item> st;
  // virtual root over the real roots
  std::vector<int> roots;
  for (int f = 0; f < nf; f++)
    if (hp.fronts[f].parent < 0) roots.push_back(f);
  auto split = [&](const std::vector<int> &kids, int a, int b, std::vector<std::array<int, 2>> &ranges) {
    // contiguous shares of [a, b) proportional to subtree work, in the children's own order
    ranges.assign(kids.size(), {a, a});
    double tot = 0;
    for (int c : kids) tot += sub[c];
    const int w = b - a;
    double acc = 0;
    int lo = a;
    for (size_t i = 0; i < kids.size(); i++) {
      acc += sub[kids[i]];
      int hi = (i + 1 == kids.size()) ? b : a + (int)(acc / std::max(tot, 1e-300) * w + 0.5);
      hi = std::min(std::max(hi, lo), b);
      ranges[i] = {lo, hi};
      lo = hi;
    }
    // a child with an empty share runs on the workgroup where its share would start (clamped into [a, b))
    for (auto &r : ranges)
      if (r[0] == r[1]) { r[0] = std::min(r[0], b - 1); r[1] = r[0] + 1; }
  };
  std::function<void(int, int, int)> visit = [&](int f, int a, int b) {
    owner[f] = a;
    std::vector<int> kids(hp.fr_child.begin() + hp.fronts[f].child0, hp.fr_child.begin() + hp.fronts[f].child0 + hp.fronts[f].child_n);
    std::vector<std::array<int, 2>> ranges;
    split(kids, a, b, ranges);
    for (size_t i = 0; i < kids.size(); i++) visit(kids[i], ranges[i][0], ranges[i][1]);
    order.push_back(f);
  };
  {
    std::vector<std::array<int, 2>> ranges;
    split(roots, 0, G, ranges);
    for (size_t i = 0; i < roots.size(); i++) visit(roots[i], ranges[i][0], ranges[i][1]);
  }
  std::vector<std::vector<uint32_t>> per(G);
  for (int f : order) per[owner[f]].push_back((uint32_t)f);
  for (int w = 0; w < G; w++) {
    first[w] = (uint32_t)list.size();
    list.insert(list.end(), per[w].begin(), per[w].end());
  }
  first[G] = (uint32_t)list.size();
}
