// exa_ropes.h — leaves and neighbour links of the rope walk, built on the host (no device code here).
//
// Input: the region kd-tree (the recursion tree of ExaBrickRegions::buildRec, exa/Regions.cpp:73-179, as ExaKdNode[]), its
// root reference, the regions' domains (6 floats each: lo, hi) and the box of the root (the union of the domains).
// Output: one leaf per region plus one per gap (an empty child slot of the tree: space no brick covers), each with its box
// and one link per face (-x +x -y +y -z +z) to whatever lies across it — an inner node (>= 0), a leaf (~index < 0), or
// kRopeOutside beyond the root box —, and the tree to descend in behind a link: the input nodes with every empty slot
// replaced by the reference of its gap leaf.
//
// One top-down pass: a node hands each child its box and its six links (the child's sibling across the split plane, the
// parent's links elsewhere), and every link is pushed down as far as it stays unambiguous — into the child next to the face
// while the linked node splits along the face's axis, or into the one child whose side of a split the whole face lies on
// (Popov, Günther, Seidel, Slusallek: "Stackless kd-tree traversal for high performance GPU ray tracing", 2007).  The top
// levels are expanded serially, the subtrees below them by a pool of threads.
//
// Used by exa_module.cpp (which adds the march's packed region record and the activity flags and uploads the result) and by
// exa_prep.cpp's diagnostic entry point exa_prep_ropes (tests/test_ropes.py checks the links on the CPU).
#pragma once
#include "../../include/exa_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

namespace exa {

enum : int32_t { kRopeOutside = EXA_KD_EMPTY + 1 };     // = the walks' "done" reference

struct RopeLeafHost {
  float   lo[3], hi[3];
  int32_t rope[6];
  int32_t region;          // region id, -1 for a gap
};

struct RopeBuild {
  std::vector<RopeLeafHost> leaves;     // [0, numRegions): the regions, then the gaps
  std::vector<ExaKdNode>    nodes;      // the tree behind the links (axis 0..2, no empty slots)
  size_t gaps = 0;
  bool   boxesMatch = true;             // the box the splits leave of every region leaf IS its domain, float for float
  bool   planesOnGrid = true;           // every plane finite, |v| <= 2^30, a multiple of 2^-10 (the short division's range)
};

// kd: nk nodes (axis in the low two bits of `axis`: the device copy keeps activity bits above them); dom: 6 floats per region
inline void buildRopesHost(const ExaKdNode *kd, size_t nk, int32_t root, const float *dom, size_t nr, const float rootLo[3],
                           const float rootHi[3], unsigned nthreads, RopeBuild &out)
{
  out.nodes.assign(kd, kd + nk);
  out.gaps = 0;
  for (ExaKdNode &n : out.nodes) {
    n.axis &= 3;
    if (n.left == EXA_KD_EMPTY) n.left = ~int32_t(nr + out.gaps++);
    if (n.right == EXA_KD_EMPTY) n.right = ~int32_t(nr + out.gaps++);
  }
  out.planesOnGrid = true;
  for (size_t i = 0; i < 6 * nr; i++) {
    const float v = dom[i];
    out.planesOnGrid = out.planesOnGrid && std::isfinite(v) && std::fabs(v) <= 1073741824.f && v * 1024.f == std::nearbyint(v * 1024.f);
  }
  out.leaves.assign(nr + out.gaps, RopeLeafHost{});
  const std::vector<ExaKdNode> &rn = out.nodes;
  struct Item { int32_t ref; float lo[3], hi[3]; int32_t rope[6]; };
  std::atomic<bool> bad{false};
  auto emitLeaf = [&](const Item &it) {
    const size_t id = size_t(~it.ref);
    RopeLeafHost &L = out.leaves[id];
    for (int k = 0; k < 3; k++) { L.lo[k] = it.lo[k]; L.hi[k] = it.hi[k]; }
    for (int f = 0; f < 6; f++) L.rope[f] = it.rope[f];
    L.region = id < nr ? (int32_t)id : -1;
    if (id < nr)
      for (int k = 0; k < 3; k++) if (dom[6 * id + k] != it.lo[k] || dom[6 * id + 3 + k] != it.hi[k]) bad = true;
  };
  // pushes the links of a box down (see above)
  auto settle = [&](Item &it) {
    for (int f = 0; f < 6; f++) {
      const int fa = f >> 1;
      const bool upper = (f & 1) != 0;
      int32_t r = it.rope[f];
      while (r >= 0) {
        const ExaKdNode &n = rn[r];
        if (n.axis == fa) r = upper ? n.left : n.right;              // the child that touches the face
        else if (n.split >= it.hi[n.axis]) r = n.left;               // the face lies on the lower side of this split
        else if (n.split <= it.lo[n.axis]) r = n.right;              // ... on the upper side
        else break;
      }
      it.rope[f] = r;
    }
  };
  // one node: its two children with their boxes and links
  auto expand = [&](const Item &it, Item &L, Item &R) {
    const ExaKdNode &n = rn[it.ref];
    L = it; R = it;
    L.ref = n.left; R.ref = n.right;
    L.hi[n.axis] = n.split; R.lo[n.axis] = n.split;
    L.rope[2 * n.axis + 1] = n.right;
    R.rope[2 * n.axis] = n.left;
    settle(L); settle(R);
  };
  auto subtree = [&](const Item &top) {
    std::vector<Item> stack(1, top);
    Item L, R;
    while (!stack.empty()) {
      const Item it = stack.back();
      stack.pop_back();
      if (it.ref < 0) { emitLeaf(it); continue; }
      expand(it, L, R);
      stack.push_back(L); stack.push_back(R);
    }
  };
  Item top{};
  top.ref = root;
  for (int k = 0; k < 3; k++) { top.lo[k] = rootLo[k]; top.hi[k] = rootHi[k]; }
  for (int f = 0; f < 6; f++) top.rope[f] = kRopeOutside;
  nthreads = nk < 4096 ? 1u : std::max(1u, nthreads);
  std::vector<Item> frontier(1, top);
  while (nthreads > 1 && frontier.size() < 64 * size_t(nthreads)) {
    std::vector<Item> next;
    bool any = false;
    for (const Item &it : frontier) {
      if (it.ref < 0) { next.push_back(it); continue; }
      Item L, R;
      expand(it, L, R);
      next.push_back(L); next.push_back(R);
      any = true;
    }
    frontier.swap(next);
    if (!any) break;
  }
  std::atomic<size_t> cursor{0};
  auto worker = [&] { for (size_t i; (i = cursor.fetch_add(1)) < frontier.size();) subtree(frontier[i]); };
  if (nthreads > 1) {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthreads; t++) pool.emplace_back(worker);
    for (auto &t : pool) t.join();
  } else worker();
  out.boxesMatch = !bad;
}

} // namespace exa
