// H-tree (Neural-Tree) construction -- SURVEY 8(f) row 3: the step BEFORE the hot path for config 4 and for the server's
// per-frame `convert_graph`.  Host code (irregular recursion over graphs of tens of nodes), no GPU involved.
//
// Replaces, procedure for procedure:
//   * generate_jth                     src/hydra_gnn/neural_tree/generate_junction_tree_hierarchies.py:26-116
//     on top of networkx.junction_tree (complete_to_chordal_graph = MCS-M, chordal_graph_cliques, maximum spanning tree of
//     the clique graph, sepset nodes) and bipartite.projected_graph (:43-47);
//   * generate_component_jth, HTree, generate_htree   src/hydra_gnn/neural_tree/construct.py:87-310
//   * the typed node / edge extraction of add_virtual_nodes_to_htree + nx_htree_to_torch   construct.py:313-371, 450-468.
// (treewidth_bound = 1000 in the reference: scene graphs never reach the sub-sampling branch; larger components are refused.)
//
// networkx leaves tie-breaking (which maximum-weight node MCS-M numbers next, the order in which cliques are reported, which of
// several equal-weight clique-graph edges Kruskal keeps) to the iteration order of Python sets, which cannot be reproduced.
// Every tie here goes to the smallest node / clique index instead.  Consequences, both tested against the reference's own
// generator (tests/test_htree_native.py): for inputs whose maximal cliques and separator incidences do not depend on ties
// (trees, complete graphs, chordal graphs in general position) the result is the reference's up to the numbering of the clique
// nodes; for all inputs it is a valid hierarchy of tree decompositions with the same structure rules.
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#include "../../include/hydra_mp.h"

namespace hmp {
char* err_buf();
}

namespace {

enum { NT_OBJECT = 0, NT_ROOM = 1, NT_OBJECT_ROOM = 2, NT_ROOM_ROOM = 3 };

struct JNode {
  bool clique = false;
  int node_type = -1;       // leaves: object / room; cliques: filled by the caller
  int has_one = -1;         // leaf: the original (global) node it copies
  std::vector<int> has;     // clique: members (global ids)
};
struct JGraph {
  std::vector<JNode> nodes;
  std::set<std::pair<int, int>> edges;  // undirected, first < second
  void add_edge(int a, int b) { if (a != b) edges.insert({std::min(a, b), std::max(a, b)}); }
  void remove_edge(int a, int b) { edges.erase({std::min(a, b), std::max(a, b)}); }
  bool has_edge(int a, int b) const { return edges.count({std::min(a, b), std::max(a, b)}) != 0; }
};

// undirected simple graph over global node ids
struct Adj {
  std::map<int, std::set<int>> nb;
  bool has(int a, int b) const { auto it = nb.find(a); return it != nb.end() && it->second.count(b) != 0; }
};

// ---- networkx.complete_to_chordal_graph (MCS-M), ties -> first in `verts` order (verts ascending) -------------------------------
bool path_through_lower(const std::vector<int>& allowed, int y, int z, const std::map<int, std::set<int>>& H) {
  // is there a path y .. z inside allowed + {y, z} ?
  std::set<int> ok(allowed.begin(), allowed.end());
  ok.insert(y); ok.insert(z);
  std::vector<int> stack{y};
  std::set<int> seen{y};
  while (!stack.empty()) {
    const int u = stack.back(); stack.pop_back();
    if (u == z) return true;
    auto it = H.find(u);
    if (it == H.end()) continue;
    for (int v : it->second)
      if (ok.count(v) && !seen.count(v)) { seen.insert(v); stack.push_back(v); }
  }
  return false;
}

// chordal completion of the subgraph induced by `verts`; returns its adjacency
std::map<int, std::set<int>> chordal_completion(const std::vector<int>& verts, const Adj& G) {
  std::map<int, std::set<int>> H;
  std::set<int> vs(verts.begin(), verts.end());
  for (int v : verts) {
    H[v];
    auto it = G.nb.find(v);
    if (it != G.nb.end())
      for (int w : it->second) if (vs.count(w)) H[v].insert(w);
  }
  const std::map<int, std::set<int>> G0 = H;  // the edge test `G.has_edge(y, z)` and the path search use the graph WITHOUT the chords
  std::map<int, int> weight;
  for (int v : verts) weight[v] = 0;
  std::vector<int> unnumbered = verts;
  std::vector<std::pair<int, int>> chords;
  for (size_t i = verts.size(); i > 0; --i) {
    size_t best = 0;
    for (size_t k = 1; k < unnumbered.size(); ++k)
      if (weight[unnumbered[k]] > weight[unnumbered[best]]) best = k;  // first maximum in list order
    const int z = unnumbered[best];
    unnumbered.erase(unnumbered.begin() + (long)best);
    std::vector<int> update;
    for (int y : unnumbered) {
      if (G0.at(y).count(z)) { update.push_back(y); continue; }
      const int yw = weight[y];
      std::vector<int> lower;
      for (int n : unnumbered) if (weight[n] < yw) lower.push_back(n);
      if (path_through_lower(lower, y, z, G0)) { update.push_back(y); chords.push_back({z, y}); }
    }
    for (int n : update) weight[n] += 1;
  }
  for (auto& c : chords) { H[c.first].insert(c.second); H[c.second].insert(c.first); }
  return H;
}

// maximal cliques of a chordal graph (maximum cardinality search; ties -> smallest id), each sorted ascending
std::vector<std::vector<int>> chordal_cliques(const std::vector<int>& verts, const std::map<int, std::set<int>>& H) {
  std::vector<std::vector<int>> out;
  std::set<int> todo(verts.begin(), verts.end());
  while (!todo.empty()) {
    // one connected component, in order of its smallest vertex
    const int start = *todo.begin();
    std::vector<int> comp, stack{start};
    std::set<int> seen{start};
    while (!stack.empty()) {
      const int u = stack.back(); stack.pop_back();
      comp.push_back(u);
      for (int v : H.at(u)) if (!seen.count(v)) { seen.insert(v); stack.push_back(v); }
    }
    for (int v : comp) todo.erase(v);
    std::sort(comp.begin(), comp.end());
    if (comp.size() == 1) { out.push_back(comp); continue; }
    std::set<int> unnumbered(comp.begin(), comp.end()), numbered;
    int v = comp[0];
    unnumbered.erase(v);
    numbered.insert(v);
    std::set<int> wanna{v};
    while (!unnumbered.empty()) {
      int best = -1, bestc = -1;
      for (int u : unnumbered) {
        int c = 0;
        for (int w : H.at(u)) if (numbered.count(w)) ++c;
        if (c > bestc) { bestc = c; best = u; }
      }
      v = best;
      unnumbered.erase(v);
      numbered.insert(v);
      std::set<int> next;
      for (int w : H.at(v)) if (numbered.count(w)) next.insert(w);
      next.insert(v);
      bool superset = true;
      for (int w : wanna) if (!next.count(w)) { superset = false; break; }
      if (!superset) out.push_back(std::vector<int>(wanna.begin(), wanna.end()));
      wanna = next;
    }
    out.push_back(std::vector<int>(wanna.begin(), wanna.end()));
  }
  return out;
}

// networkx.junction_tree + bipartite.projected_graph onto the cliques: the cliques and which pairs of them end up adjacent
void junction_tree_cliques(const std::vector<int>& verts, const Adj& G, std::vector<std::vector<int>>& cliques,
                           std::set<std::pair<int, int>>& clique_edges) {
  const auto H = chordal_completion(verts, G);
  cliques = chordal_cliques(verts, H);
  const int K = (int)cliques.size();
  struct E { int w, a, b; std::vector<int> sep; };
  std::vector<E> es;
  for (int a = 0; a < K; ++a)
    for (int b = a + 1; b < K; ++b) {
      std::vector<int> sep;
      std::set_intersection(cliques[a].begin(), cliques[a].end(), cliques[b].begin(), cliques[b].end(), std::back_inserter(sep));
      if (!sep.empty()) es.push_back({(int)sep.size(), a, b, sep});
    }
  std::stable_sort(es.begin(), es.end(), [](const E& x, const E& y) { return x.w > y.w; });  // Kruskal, maximum weight first
  std::vector<int> parent(K);
  for (int i = 0; i < K; ++i) parent[i] = i;
  auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
  // one sepset node per distinct separator; the projection joins every pair of cliques hanging off the same sepset node
  std::map<std::vector<int>, std::set<int>> sep_members;
  for (const E& e : es) {
    const int ra = find(e.a), rb = find(e.b);
    if (ra == rb) continue;
    parent[ra] = rb;
    sep_members[e.sep].insert(e.a);
    sep_members[e.sep].insert(e.b);
  }
  clique_edges.clear();
  for (auto& kv : sep_members) {
    std::vector<int> m(kv.second.begin(), kv.second.end());
    for (size_t i = 0; i < m.size(); ++i)
      for (size_t j = i + 1; j < m.size(); ++j) clique_edges.insert({m[i], m[j]});
  }
}

// ---- generate_jth (generate_junction_tree_hierarchies.py:26-116), remove_edges_every_layer = True -----------------------------------
struct JthResult {
  JGraph g;
  std::vector<int> roots;  // empty + single == true: the one-node original graph
  bool single = false;
};

JGraph leaves_only(const std::vector<int>& verts, const std::vector<int>& node_type_of) {
  JGraph g;
  for (int v : verts) {
    JNode n;
    n.clique = false; n.has_one = v; n.node_type = node_type_of[v];
    g.nodes.push_back(n);
  }
  return g;
}

JthResult generate_jth(const std::vector<int>& verts, const Adj& G, bool original, const std::vector<int>& node_type_of) {
  JthResult R;
  if (verts.size() == 1 && original) {
    R.g = leaves_only(verts, node_type_of);
    R.single = true;
    return R;
  }
  std::vector<std::vector<int>> cliques;
  std::set<std::pair<int, int>> cedges;
  junction_tree_cliques(verts, G, cliques, cedges);
  const int K = (int)cliques.size();
  if (K == 1 && !original) {  // a clique of the parent level: its members hang off the parent directly
    R.g = leaves_only(verts, node_type_of);
    for (int i = 0; i < (int)verts.size(); ++i) R.roots.push_back(i);
    return R;
  }
  JGraph& J = R.g;
  for (int c = 0; c < K; ++c) {
    JNode n;
    n.clique = true; n.has = cliques[c];
    J.nodes.push_back(n);
    R.roots.push_back(c);
  }
  for (auto& e : cedges) J.add_edge(e.first, e.second);
  for (int a = 0; a < K; ++a) {
    const std::vector<int>& mem = cliques[a];
    if (mem.size() == 1) continue;
    if (mem.size() == 2) {
      for (int v : mem) {
        JNode n;
        n.clique = false; n.has_one = v; n.node_type = node_type_of[v];
        J.nodes.push_back(n);
        J.add_edge((int)J.nodes.size() - 1, a);
      }
      continue;
    }
    JthResult sub = generate_jth(mem, G, false, node_type_of);
    // the tree among the sub-level's top nodes is dropped (remove_edges_every_layer)
    for (size_t i = 0; i < sub.roots.size(); ++i)
      for (size_t j = i + 1; j < sub.roots.size(); ++j) sub.g.remove_edge(sub.roots[i], sub.roots[j]);
    const int base = (int)J.nodes.size();
    for (auto& n : sub.g.nodes) J.nodes.push_back(n);
    for (auto& e : sub.g.edges) J.add_edge(base + e.first, base + e.second);
    for (int r : sub.roots) J.add_edge(base + r, a);
  }
  return R;
}

}  // namespace

struct hmp_htree {
  int32_t counts[4] = {0, 0, 0, 0};
  std::vector<int32_t> object_orig, room_orig;
  std::vector<int32_t> edges[10];  // [2][n] each: row 0 sources, row 1 destinations (local indices inside the node types)
  std::vector<int32_t> init[3];    // ov_to_or, rv_to_or, rv_to_rr: row 0 = virtual (original index inside its type), row 1 = clique
};

namespace {

const int ET_SRC[10] = {NT_OBJECT, NT_OBJECT_ROOM, NT_ROOM, NT_OBJECT_ROOM, NT_ROOM, NT_ROOM_ROOM, NT_OBJECT_ROOM, NT_ROOM_ROOM, NT_OBJECT_ROOM, NT_ROOM_ROOM};
const int ET_DST[10] = {NT_OBJECT_ROOM, NT_OBJECT, NT_OBJECT_ROOM, NT_ROOM, NT_ROOM_ROOM, NT_ROOM, NT_ROOM_ROOM, NT_OBJECT_ROOM, NT_OBJECT_ROOM, NT_ROOM_ROOM};

void push2(std::vector<int32_t>& rows, std::vector<int32_t>& tmp_dst, int32_t s, int32_t d) { rows.push_back(s); tmp_dst.push_back(d); }

}  // namespace

extern "C" int hmp_htree_build(int32_t n_objects, int32_t n_rooms, const int64_t* oo, int64_t e_oo, const int64_t* rr, int64_t e_rr,
                               const int64_t* ro, int64_t e_ro, hmp_htree** out) {
  auto fail = [](const char* m) { snprintf(hmp::err_buf(), 512, "hmp_htree_build: %s", m); return HMP_E_ARG; };
  if (!out || n_objects < 0 || n_rooms < 0 || e_oo < 0 || e_rr < 0 || e_ro < 0) return fail("bad argument");
  if ((e_oo && !oo) || (e_rr && !rr) || (e_ro && !ro)) return fail("null edge list");
  const int N = n_objects + n_rooms;  // global ids: objects first, then rooms (construct.py:49-60 via to_homogeneous)
  std::vector<int> node_type_of(N);
  for (int i = 0; i < N; ++i) node_type_of[i] = i < n_objects ? NT_OBJECT : NT_ROOM;
  Adj G;
  for (int i = 0; i < N; ++i) G.nb[i];
  auto add = [&](int64_t a, int64_t b, int64_t na, int64_t nb_, int off_a, int off_b) -> bool {
    if (a < 0 || a >= na || b < 0 || b >= nb_) return false;
    const int u = (int)a + off_a, v = (int)b + off_b;
    if (u != v) { G.nb[u].insert(v); G.nb[v].insert(u); }
    return true;
  };
  for (int64_t k = 0; k < e_oo; ++k) if (!add(oo[k], oo[e_oo + k], n_objects, n_objects, 0, 0)) return fail("object edge endpoint out of range");
  for (int64_t k = 0; k < e_rr; ++k) if (!add(rr[k], rr[e_rr + k], n_rooms, n_rooms, n_objects, n_objects)) return fail("room edge endpoint out of range");
  for (int64_t k = 0; k < e_ro; ++k) if (!add(ro[k], ro[e_ro + k], n_rooms, n_objects, n_objects, 0)) return fail("room-object edge endpoint out of range");

  // the object / room graphs seen by the decompositions hold edges of ONE layer only
  Adj Goo, Grr;
  for (int i = 0; i < N; ++i) {
    for (int v : G.nb[i]) {
      if (i < n_objects && v < n_objects) Goo.nb[i].insert(v);
      if (i >= n_objects && v >= n_objects) Grr.nb[i].insert(v);
    }
    Goo.nb[i]; Grr.nb[i];
  }

  JGraph H;  // the whole H-tree (disjoint union over connected components)
  std::vector<bool> seen(N, false);
  for (int s0 = 0; s0 < N; ++s0) {
    if (seen[s0]) continue;
    std::vector<int> comp, stack{s0};
    seen[s0] = true;
    while (!stack.empty()) {
      const int u = stack.back(); stack.pop_back();
      comp.push_back(u);
      for (int v : G.nb[u]) if (!seen[v]) { seen[v] = true; stack.push_back(v); }
    }
    std::sort(comp.begin(), comp.end());
    std::vector<int> rooms;
    for (int v : comp) if (v >= n_objects) rooms.push_back(v);
    if (rooms.empty()) return fail("a connected component without a room node (the reference's generate_htree needs one)");
    if (comp.size() > 1000) return fail("component with more than 1000 nodes: the reference sub-samples here (treewidth_bound), not implemented");
    // ---- room tree (generate_component_jth, component_type "rooms")
    JthResult RJ = generate_jth(rooms, Grr, true, node_type_of);
    if (RJ.single) RJ.roots = {0};
    for (auto& n : RJ.g.nodes) if (n.clique) n.node_type = NT_ROOM_ROOM;
    JGraph jth = RJ.g;
    const std::vector<int> room_roots = RJ.roots;
    // ---- per room, per connected object component: the object tree with a copy of the room on its lowest cliques
    for (int r : rooms) {
      std::vector<int> objs;
      for (int v : G.nb[r]) if (v < n_objects) objs.push_back(v);
      std::set<int> objset(objs.begin(), objs.end()), done;
      for (int o0 : objs) {
        if (done.count(o0)) continue;
        std::vector<int> oc, st{o0};
        done.insert(o0);
        while (!st.empty()) {
          const int u = st.back(); st.pop_back();
          oc.push_back(u);
          for (int v : Goo.nb[u]) if (objset.count(v) && !done.count(v)) { done.insert(v); st.push_back(v); }
        }
        std::sort(oc.begin(), oc.end());
        // the component graph is `dsg_nx.subgraph(oc)`: object edges among the component's nodes
        JthResult OJ = generate_jth(oc, Goo, true, node_type_of);
        JGraph oj = OJ.g;
        std::vector<int> oroots = OJ.roots;
        if (OJ.single) {  // one object: a clique of its own above it (construct.py:145-155)
          JNode c;
          c.clique = true; c.has = {oj.nodes[0].has_one};
          oj.nodes.push_back(c);
          oj.add_edge(0, 1);
          oroots = {1};
        }
        for (auto& n : oj.nodes) if (n.clique) { n.node_type = NT_OBJECT_ROOM; n.has.push_back(r); }
        std::set<int> lowest;
        for (auto& e : oj.edges) {
          if (!oj.nodes[e.first].clique && oj.nodes[e.second].clique) lowest.insert(e.second);
          if (!oj.nodes[e.second].clique && oj.nodes[e.first].clique) lowest.insert(e.first);
        }
        for (int c : lowest) {
          JNode n;
          n.clique = false; n.has_one = r; n.node_type = NT_ROOM;
          oj.nodes.push_back(n);
          oj.add_edge(c, (int)oj.nodes.size() - 1);
        }
        // HTree.add_object_jth (construct.py:216-236)
        for (size_t i = 0; i < oroots.size(); ++i)
          for (size_t j = i + 1; j < oroots.size(); ++j) oj.remove_edge(oroots[i], oroots[j]);
        for (int root : room_roots) {
          const JNode& rn = jth.nodes[root];
          const bool holds = rn.clique ? (std::find(rn.has.begin(), rn.has.end(), r) != rn.has.end()) : (rn.has_one == r);
          if (!holds) continue;
          const int base = (int)jth.nodes.size();
          for (auto& n : oj.nodes) jth.nodes.push_back(n);
          for (auto& e : oj.edges) jth.add_edge(base + e.first, base + e.second);
          for (int orr : oroots) jth.add_edge(root, base + orr);
        }
      }
    }
    const int base = (int)H.nodes.size();
    for (auto& n : jth.nodes) H.nodes.push_back(n);
    for (auto& e : jth.edges) H.add_edge(base + e.first, base + e.second);
  }

  // ---- typed arrays (construct.py:313-371, 450-468) ---------------------------------------------------------------------------------------
  hmp_htree* T = new hmp_htree;
  std::vector<int> local(H.nodes.size());
  for (size_t i = 0; i < H.nodes.size(); ++i) {
    const JNode& n = H.nodes[i];
    local[i] = T->counts[n.node_type]++;
    if (n.node_type == NT_OBJECT) T->object_orig.push_back(n.has_one);
    if (n.node_type == NT_ROOM) T->room_orig.push_back(n.has_one - n_objects);
  }
  std::vector<int32_t> dst[10];
  std::vector<std::vector<int>> nbrs(H.nodes.size());
  for (auto& e : H.edges) { nbrs[e.first].push_back(e.second); nbrs[e.second].push_back(e.first); }
  for (size_t u = 0; u < H.nodes.size(); ++u) {
    std::sort(nbrs[u].begin(), nbrs[u].end());
    for (int v : nbrs[u]) {
      const int tu = H.nodes[u].node_type, tv = H.nodes[v].node_type;
      int et = -1;
      for (int k = 0; k < 10; ++k) if (ET_SRC[k] == tu && ET_DST[k] == tv) et = k;
      if (et < 0) { delete T; return fail("internal: an edge between node types the H-tree schema does not know"); }
      push2(T->edges[et], dst[et], local[u], local[v]);
    }
  }
  for (int k = 0; k < 10; ++k) T->edges[k].insert(T->edges[k].end(), dst[k].begin(), dst[k].end());
  std::vector<int32_t> idst[3];
  for (size_t i = 0; i < H.nodes.size(); ++i) {
    const JNode& n = H.nodes[i];
    if (!n.clique) continue;
    for (int m : n.has) {
      int which;
      if (n.node_type == NT_OBJECT_ROOM) which = m < n_objects ? 0 : 1;
      else which = 2;
      if (n.node_type == NT_ROOM_ROOM && m < n_objects) { delete T; return fail("internal: object inside a room clique"); }
      push2(T->init[which], idst[which], m < n_objects ? m : m - n_objects, local[i]);
    }
  }
  for (int k = 0; k < 3; ++k) T->init[k].insert(T->init[k].end(), idst[k].begin(), idst[k].end());
  *out = T;
  return HMP_OK;
}

extern "C" int hmp_htree_sizes(const hmp_htree* t, int32_t* counts4, int64_t* n_edges10, int64_t* n_init3) {
  if (!t || !counts4 || !n_edges10 || !n_init3) { snprintf(hmp::err_buf(), 512, "hmp_htree_sizes: null argument"); return HMP_E_ARG; }
  for (int k = 0; k < 4; ++k) counts4[k] = t->counts[k];
  for (int k = 0; k < 10; ++k) n_edges10[k] = (int64_t)t->edges[k].size() / 2;
  for (int k = 0; k < 3; ++k) n_init3[k] = (int64_t)t->init[k].size() / 2;
  return HMP_OK;
}

extern "C" int hmp_htree_fill(const hmp_htree* t, int32_t* object_orig, int32_t* room_orig, int32_t* const* edges10, int32_t* const* init3) {
  if (!t || !edges10 || !init3) { snprintf(hmp::err_buf(), 512, "hmp_htree_fill: null argument"); return HMP_E_ARG; }
  if (object_orig && !t->object_orig.empty()) memcpy(object_orig, t->object_orig.data(), t->object_orig.size() * 4);
  if (room_orig && !t->room_orig.empty()) memcpy(room_orig, t->room_orig.data(), t->room_orig.size() * 4);
  for (int k = 0; k < 10; ++k)
    if (edges10[k] && !t->edges[k].empty()) memcpy(edges10[k], t->edges[k].data(), t->edges[k].size() * 4);
  for (int k = 0; k < 3; ++k)
    if (init3[k] && !t->init[k].empty()) memcpy(init3[k], t->init[k].data(), t->init[k].size() * 4);
  return HMP_OK;
}

extern "C" void hmp_htree_destroy(hmp_htree* t) { delete t; }
