// Structural validation of a serialised MemoryBlock's node array (Octree::ToMemoryBlock layout, Octree.cpp:424-456).
//
// The reference's FromMemoryBlock (Octree.cpp:403-421) trusts its input; the entry points here take bytes from
// anywhere (files, other processes), so both deserialisers -- hpsdf_tree_upload and the continuity post-process --
// walk the tree from the root with these checks before anything indexes by what the block says:
//   * an interior node's 8 children lie inside the node array (compared without wrap-around) and behind the root;
//   * every node is reached at most once (no cycles, no shared sub-trees): the walk terminates in <= nNodes steps;
//   * the path length never exceeds TREE_MAX_DEPTH + 1 and equals the leaf's stored depth;
//   * a leaf's degree is <= BASIS_MAX_DEGREE and its coefficient range lies inside the store (no wrap-around);
//   * (optional) the leaves' coefficient ranges are pairwise disjoint -- the continuity assembly writes one row
//     block per leaf in parallel.
// A leaf is a node whose degree != 13, as Octree::Query decides (Octree.cpp:686); `strictLeafMarker` also insists on
// childIdx == all-ones for leaves and != all-ones for interior nodes (what ToMemoryBlock writes).
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../include/hpsdf.h"

namespace hpsdf {

struct BlockTreeInfo {
    std::vector<uint8_t> depthOf;   // per node: path length from the root (reached nodes only)
    std::vector<uint8_t> reached;   // per node: 1 if the walk from the root reaches it
    std::vector<uint64_t> order;    // nodes in the order the walk pops them (a stack walk; children pushed 0..7)
    uint64_t leaves = 0;
    int maxDegree = 0, maxDepth = 0, minLeafDepth = HPSDF_TREE_MAX_DEPTH + 1;
};

inline int checkBlockTree(const hpsdf_node* nodes, uint64_t nNodes, uint64_t nCoeffs, const uint64_t* coeffCount,
                          bool strictLeafMarker, bool disjointLeaves, BlockTreeInfo* info, std::string& err) {
    const uint64_t kLeaf = ~0ull;
    if (nNodes == 0) {
        err = "empty node array";
        return HPSDF_ERR_BAD_BLOCK;
    }
    BlockTreeInfo local;
    BlockTreeInfo& I = info ? *info : local;
    I.depthOf.assign(nNodes, 0);
    I.reached.assign(nNodes, 0);
    I.order.clear();
    I.order.reserve(nNodes);
    std::vector<uint64_t> stack{0};
    I.reached[0] = 1;
    std::vector<std::pair<uint64_t, uint64_t>> ranges;
    while (!stack.empty()) {
        const uint64_t i = stack.back();
        stack.pop_back();
        I.order.push_back(i);
        const hpsdf_node& n = nodes[i];
        if (n.degree == HPSDF_INTERIOR_DEGREE) {
            if (strictLeafMarker && n.child_idx == kLeaf) {
                err = "interior node without children";
                return HPSDF_ERR_BAD_BLOCK;
            }
            // children c .. c+7 must exist: c < nNodes and nNodes - c >= 8 (no wrap-around), and never the root
            if (n.child_idx == 0 || n.child_idx >= nNodes || nNodes - n.child_idx < 8) {
                err = "child index out of range";
                return HPSDF_ERR_BAD_BLOCK;
            }
            if (I.depthOf[i] >= HPSDF_TREE_MAX_DEPTH + 1) {
                err = "tree deeper than TREE_MAX_DEPTH + 1";
                return HPSDF_ERR_BAD_BLOCK;
            }
            for (unsigned c = 0; c < 8; ++c) {
                const uint64_t ch = n.child_idx + c;
                if (I.reached[ch]) {
                    err = "node reached twice (cycle or shared children)";
                    return HPSDF_ERR_BAD_BLOCK;
                }
                I.reached[ch] = 1;
                I.depthOf[ch] = (uint8_t)(I.depthOf[i] + 1);
            }
            for (unsigned c = 0; c < 8; ++c) stack.push_back(n.child_idx + c);
        } else {
            if (strictLeafMarker && n.child_idx != kLeaf) {
                err = "leaf with a child index";
                return HPSDF_ERR_BAD_BLOCK;
            }
            if (n.degree > HPSDF_BASIS_MAX_DEGREE) {
                err = "leaf degree out of range";
                return HPSDF_ERR_BAD_BLOCK;
            }
            const uint64_t cc = coeffCount[n.degree];
            if (n.coeffs_start > nCoeffs || cc > nCoeffs - n.coeffs_start) {
                err = "leaf coefficients out of range";
                return HPSDF_ERR_BAD_BLOCK;
            }
            if (n.depth != I.depthOf[i]) {
                err = "stored depth does not match tree depth";
                return HPSDF_ERR_BAD_BLOCK;
            }
            if (disjointLeaves) ranges.emplace_back(n.coeffs_start, n.coeffs_start + cc);
            ++I.leaves;
            I.maxDegree = std::max(I.maxDegree, (int)n.degree);
            I.maxDepth = std::max(I.maxDepth, (int)I.depthOf[i]);
            I.minLeafDepth = std::min(I.minLeafDepth, (int)I.depthOf[i]);
        }
    }
    if (disjointLeaves) {
        std::sort(ranges.begin(), ranges.end());
        for (size_t k = 1; k < ranges.size(); ++k)
            if (ranges[k].first < ranges[k - 1].second) {
                err = "leaf coefficient ranges overlap";
                return HPSDF_ERR_BAD_BLOCK;
            }
    }
    return HPSDF_OK;
}

}  // namespace hpsdf
