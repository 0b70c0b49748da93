// nanoflann's visit order for tied distances — shared by normals.hip (device comparator) and the host-side builder.
// See the header comment of normals.hip ("EXACT TIES").  Plain C++ apart from the MV_HD marker, so the builder and the comparator can
// also be exercised by a host-only harness.
#pragma once
#include <algorithm>
#include <vector>

#if defined(__HIPCC__)
#define MV_HD __host__ __device__
#else
#define MV_HD
#endif

namespace mvicp {

// Node i covers the slots [first, last) of the tree's point ordering; inner nodes: children = i + 1 (slots < split) and `right`
// (slots >= split); lo_cut / hi_cut = the children's extents along `axis` at the split (divlow / divhigh, nanoflann.hpp:1070-1071).
struct VisitNode { int split; int right; int axis; int pad; double lo_cut, hi_cut; };
struct VisitTree { const VisitNode* nodes; const int* slot; };   // slot[original index] = position of the point in the tree's ordering

// true iff the tree's exact search for `q` reaches point a (original index) before point b: descend to their lowest common ancestor;
// there the near child is the low one iff (q[axis] - lo_cut) + (q[axis] - hi_cut) < 0 (nanoflann.hpp:1222-1233).
MV_HD inline bool visited_before(const VisitTree& T, double qx, double qy, double qz, long long a, long long b) {
  const int sa = T.slot[a], sb = T.slot[b];
  int i = 0;
  for (;;) {
    const VisitNode nd = T.nodes[i];
    if (sa < nd.split && sb < nd.split) { i = i + 1; continue; }
    if (sa >= nd.split && sb >= nd.split) { i = nd.right; continue; }
    const double v = nd.axis == 0 ? qx : nd.axis == 1 ? qy : qz;
    const bool low_first = (v - nd.lo_cut) + (v - nd.hi_cut) < 0.0;
    return (sa < nd.split) == low_first;
  }
}


// Host: the split structure nanoflann's KDTreeSingleIndexAdaptor builds for this cloud with leaf_max_size = 1 (buildIndex ->
// divideTree -> middleSplit_ -> planeSplit, nanoflann.hpp:859-867,1034-1174), restated — only the topology and the cut values are
// kept, which is all the visit order depends on.  Points are handled through an index permutation `ord` that starts as 0..n-1.
namespace kdv_detail {   // (a named namespace: the inline builder below has external linkage and must see the same types in every translation unit)

struct Box3 { double lo[3], hi[3]; };

struct VisitBuilder {
  const double* xyz; std::vector<int> ord; std::vector<VisitNode> nodes;
  double at(int slot, int axis) const { return xyz[3 * (size_t)ord[slot] + axis]; }
  void range_of(int first, int count, int axis, double& mn, double& mx) const {
    mn = mx = at(first, axis);
    for (int i = 1; i < count; ++i) { const double v = at(first + i, axis); if (v < mn) mn = v; if (v > mx) mx = v; }
  }
  // One sweep of the two-ended exchange partition: afterwards every slot before the returned position holds a value for which
  // `goes_low` is true.  `start` = where the low end begins (the second sweep continues where the first stopped).
  template <class Low> int sweep(int first, int count, int axis, int start, Low goes_low) {
    long long lo = start, hi = (long long)count - 1;
    for (;;) {
      while (lo <= hi && goes_low(at(first + (int)lo, axis))) ++lo;
      while (hi != 0 && lo <= hi && !goes_low(at(first + (int)hi, axis))) --hi;
      if (lo > hi || hi == 0) break;
      std::swap(ord[first + (int)lo], ord[first + (int)hi]);
      ++lo; --hi;
    }
    return (int)lo;
  }

  // The split of one range: picks axis / cut value, partitions the slots, returns the split position k (first < first + k < last).
  int split_range(int first, int last, const Box3& box, int& axis_out, double& cut_out) {
    const int count = last - first;
    // cut axis: among the axes whose box extent is within 1e-5 of the largest, the one with the largest spread of the points — where
    // the spread is measured along the axis chosen SO FAR (nanoflann.hpp:1109 passes `cutfeat`, not `i`); kept as it is
    double span_max = box.hi[0] - box.lo[0];
    for (int a = 1; a < 3; ++a) span_max = std::max(span_max, box.hi[a] - box.lo[a]);
    int axis = 0;
    double best_spread = -1.0;
    for (int a = 0; a < 3; ++a)
      if (box.hi[a] - box.lo[a] > (1.0 - 0.00001) * span_max) {
        double mn, mx;
        range_of(first, count, axis, mn, mx);
        if (mx - mn > best_spread) { axis = a; best_spread = mx - mn; }
      }
    // cut value: the middle of the box, pulled into the range of the points
    double mn, mx;
    range_of(first, count, axis, mn, mx);
    const double mid = (box.lo[axis] + box.hi[axis]) / 2;
    const double cut = mid < mn ? mn : (mid > mx ? mx : mid);
    // slots [0, below) < cut, [below, upto) == cut, [upto, count) > cut; the split position is the middle if it falls inside the run of
    // equal values, else the nearer end of that run
    const int below = sweep(first, count, axis, 0, [cut](double v) { return v < cut; });
    const int upto = sweep(first, count, axis, below, [cut](double v) { return v <= cut; });
    const int half = count / 2;
    axis_out = axis; cut_out = cut;
    return below > half ? below : (upto < half ? upto : half);
  }

  // divideTree (nanoflann.hpp:1034-1078) with an EXPLICIT stack: nanoflann recurses once per level, and a degenerate cloud (coordinates in
  // geometric progression: every middle split peels off one point) is as deep as the exponent range allows — the reference would walk
  // its call stack that deep; this builder does not depend on it.  Nodes are numbered in pre-order (left child = parent + 1), boxes are
  // merged bottom-up when a node's right subtree completes, exactly as the recursive form did.
  int divide(int first0, int last0, Box3& box0) {
    struct Frame { int me, first, last, k, axis, stage; Box3 box, lb, rb; };   // stage 0: enter, 1: left done, 2: right done
    std::vector<Frame> st;
    Box3 ret = box0;      // box handed back by the subtree that has just completed
    st.push_back(Frame{-1, first0, last0, 0, 0, 0, box0, box0, box0});
    int root = -1;
    while (!st.empty()) {
      Frame& f = st.back();
      if (f.stage == 0) {
        f.me = (int)nodes.size();
        if (root < 0) root = f.me;
        nodes.push_back(VisitNode{f.last, -1, -1, 0, 0.0, 0.0});
        const int count = f.last - f.first;
        if (count <= 1) {                                       // leaf (leaf_max_size = 1): its box is the point itself
          for (int a = 0; a < 3; ++a) ret.lo[a] = ret.hi[a] = count > 0 ? at(f.first, a) : 0.0;
          st.pop_back();
          continue;
        }
        double cut;
        f.k = split_range(f.first, f.last, f.box, f.axis, cut);
        f.lb = f.box; f.rb = f.box;
        f.lb.hi[f.axis] = cut; f.rb.lo[f.axis] = cut;
        f.stage = 1;
        const Frame child{-1, f.first, f.first + f.k, 0, 0, 0, f.lb, f.lb, f.lb};
        st.push_back(child);                                    // (f is dangling from here on: re-fetched at the top of the loop)
        continue;
      }
      if (f.stage == 1) {
        f.lb = ret;                                             // the left subtree's tightened box
        f.stage = 2;
        nodes[f.me].right = (int)nodes.size();                  // pre-order: the right child is the next node to be created
        const Frame child{-1, f.first + f.k, f.last, 0, 0, 0, f.rb, f.rb, f.rb};
        st.push_back(child);
        continue;
      }
      f.rb = ret;
      VisitNode& nd = nodes[f.me];
      nd.split = f.first + f.k; nd.axis = f.axis; nd.lo_cut = f.lb.hi[f.axis]; nd.hi_cut = f.rb.lo[f.axis];
      for (int a = 0; a < 3; ++a) { ret.lo[a] = std::min(f.lb.lo[a], f.rb.lo[a]); ret.hi[a] = std::max(f.lb.hi[a], f.rb.hi[a]); }
      st.pop_back();
    }
    box0 = ret;
    return root;
  }
};

}  // namespace kdv_detail

inline int build_visit_tree(const double* xyz, int n, std::vector<VisitNode>& nodes, std::vector<int>& slot) {
  kdv_detail::VisitBuilder B;
  B.xyz = xyz; B.ord.resize(n);
  for (int i = 0; i < n; ++i) B.ord[i] = i;
  B.nodes.reserve(2 * (size_t)n);
  kdv_detail::Box3 box;
  for (int a = 0; a < 3; ++a) box.lo[a] = box.hi[a] = xyz[a];
  for (int i = 1; i < n; ++i)
    for (int a = 0; a < 3; ++a) { const double v = xyz[3 * (size_t)i + a]; if (v < box.lo[a]) box.lo[a] = v; if (v > box.hi[a]) box.hi[a] = v; }
  B.divide(0, n, box);
  nodes.swap(B.nodes);
  slot.assign(n, 0);
  for (int s2 = 0; s2 < n; ++s2) slot[B.ord[s2]] = s2;
  return 0;
}


}  // namespace mvicp
