// quadtree.h -- host-side keypoint thinning ("OctTree") for the MI355X ORB front end.
//
// Behaviour contract: ORBextractor::DistributeOctTree + ExtractorNode::DivideNode of the reference
// (src/ORBextractor.cc:513-569, 571-795): same node boxes, same list order (children pushed to the
// front in n1..n4 order, parents erased), same stop rules, same "largest nodes first" final phase,
// same "highest response, first wins" pick.  Ties between equal-sized nodes in the final phase are
// broken by creation order (later-created first), the documented stand-in for the reference's
// pointer-value tie-break.
//
// Implementation is NOT the reference's: no std::list of nodes that own copies of cv::KeyPoint
// vectors.  Candidates stay in flat SoA arrays; a node owns a [begin,end) range of candidate
// indices in one of two ping-pong index buffers (a 4-way stable partition moves a parent's range
// into the other buffer, split into the children's sub-ranges), and the "list" is an intrusive
// doubly linked list over a node pool.  O(n * depth) index moves, zero per-node allocations.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace orbfe {

struct QuadTree {
  struct NodeRec {
    int x0, x1, y0, y1;  // UL.x, UR.x, UL.y, BL.y (the four corners always form this rectangle)
    int begin, end;      // candidate index range in ibuf[buf]
    int prev, next;      // list links (-1 = none)
    int seq;             // creation order
    uint8_t buf;
    bool noMore;
  };
  std::vector<NodeRec> pool;
  std::vector<int> ibuf[2];
  std::vector<std::pair<int, int>> toExpand, prevToExpand;  // (size, node id)
  int head = -1, tail = -1, count = 0, seq = 0;

  int newNode() {
    pool.emplace_back();
    return (int)pool.size() - 1;
  }
  void pushFront(int id) {
    NodeRec& n = pool[id];
    n.prev = -1;
    n.next = head;
    if (head >= 0) pool[head].prev = id; else tail = id;
    head = id;
    count++;
  }
  void pushBack(int id) {
    NodeRec& n = pool[id];
    n.next = -1;
    n.prev = tail;
    if (tail >= 0) pool[tail].next = id; else head = id;
    tail = id;
    count++;
  }
  int erase(int id) {  // returns the next node
    NodeRec& n = pool[id];
    const int nx = n.next;
    if (n.prev >= 0) pool[n.prev].next = n.next; else head = n.next;
    if (n.next >= 0) pool[n.next].prev = n.prev; else tail = n.prev;
    count--;
    return nx;
  }

  // Split node `id` (DivideNode), push the non-empty children to the front (n1..n4 order), record
  // those with more than one candidate.  x/y: candidate coordinates relative to (minX, minY).
  void divideAndPush(int id, const int16_t* x, const int16_t* y, int& nToExpand) {
    const NodeRec p = pool[id];  // copy: pool may reallocate below
    const int halfX = (int)std::ceil(static_cast<float>(p.x1 - p.x0) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(p.y1 - p.y0) / 2);
    const int midX = p.x0 + halfX, midY = p.y0 + halfY;
    const int* src = ibuf[p.buf].data();
    int* dst = ibuf[p.buf ^ 1].data();
    int cnt[4] = {0, 0, 0, 0};
    for (int i = p.begin; i < p.end; i++) {
      const int k = src[i];
      cnt[(x[k] < midX ? 0 : 1) + (y[k] < midY ? 0 : 2)]++;
    }
    int pos[4];
    pos[0] = p.begin;
    pos[1] = pos[0] + cnt[0];
    pos[2] = pos[1] + cnt[1];
    pos[3] = pos[2] + cnt[2];
    const int start[4] = {pos[0], pos[1], pos[2], pos[3]};
    for (int i = p.begin; i < p.end; i++) {
      const int k = src[i];
      dst[pos[(x[k] < midX ? 0 : 1) + (y[k] < midY ? 0 : 2)]++] = k;
    }
    const int bx0[4] = {p.x0, midX, p.x0, midX}, bx1[4] = {midX, p.x1, midX, p.x1};
    const int by0[4] = {p.y0, p.y0, midY, midY}, by1[4] = {midY, midY, p.y1, p.y1};
    for (int c = 0; c < 4; c++) {
      if (cnt[c] == 0) continue;
      const int nid = newNode();
      NodeRec& n = pool[nid];
      n.x0 = bx0[c]; n.x1 = bx1[c]; n.y0 = by0[c]; n.y1 = by1[c];
      n.begin = start[c]; n.end = start[c] + cnt[c];
      n.buf = p.buf ^ 1;
      n.noMore = (cnt[c] == 1);
      n.seq = seq++;
      pushFront(nid);
      if (cnt[c] > 1) {
        nToExpand++;
        toExpand.emplace_back(cnt[c], nid);
      }
    }
  }

  // Returns the indices (into the candidate arrays) of the retained keypoints, in list order.
  // x, y: coordinates relative to (minX, minY); score: FAST response.
  void distribute(const int16_t* x, const int16_t* y, const uint8_t* score, int n, int minX, int maxX,
                  int minY, int maxY, int N, std::vector<int>& out) {
    out.clear();
    pool.clear();
    toExpand.clear();
    head = tail = -1;
    count = 0;
    seq = 0;
    if (n <= 0) return;
    ibuf[0].resize(n);
    ibuf[1].resize(n);
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni <= 0) return;  // the reference divides by zero here; callers reject such sizes earlier
    const float hX = static_cast<float>(maxX - minX) / nIni;
    pool.reserve(4 * (size_t)std::max(N, 16) + 64);
    // root assignment keeps candidate order inside each root: counting sort by root
    std::vector<int> rootCnt(nIni + 1, 0);
    std::vector<int> rootOf(n);
    for (int i = 0; i < n; i++) {
      int r = (int)((float)x[i] / hX);
      if (r >= nIni) r = nIni - 1;  // unreachable for in-range x; keeps indexing safe
      rootOf[i] = r;
      rootCnt[r + 1]++;
    }
    for (int r = 0; r < nIni; r++) rootCnt[r + 1] += rootCnt[r];
    {
      std::vector<int> pos(rootCnt.begin(), rootCnt.end() - 1);
      for (int i = 0; i < n; i++) ibuf[0][pos[rootOf[i]]++] = i;
    }
    for (int i = 0; i < nIni; i++) {
      const int id = newNode();
      NodeRec& r = pool[id];
      r.x0 = (int)(hX * static_cast<float>(i));
      r.x1 = (int)(hX * static_cast<float>(i + 1));
      r.y0 = 0;
      r.y1 = maxY - minY;
      r.begin = rootCnt[i];
      r.end = rootCnt[i + 1];
      r.buf = 0;
      r.noMore = false;
      r.seq = seq++;
      pushBack(id);
    }
    for (int id = head; id >= 0;) {
      NodeRec& r = pool[id];
      const int sz = r.end - r.begin;
      if (sz == 1) { r.noMore = true; id = r.next; }
      else if (sz == 0) id = erase(id);
      else id = r.next;
    }

    bool finish = false;
    while (!finish) {
      int prevSize = count;
      int nToExpand = 0;
      toExpand.clear();
      for (int id = head; id >= 0;) {
        if (pool[id].noMore) { id = pool[id].next; continue; }
        divideAndPush(id, x, y, nToExpand);
        id = erase(id);
      }
      if (count >= N || count == prevSize) {
        finish = true;
      } else if (count + nToExpand * 3 > N) {
        while (!finish) {
          prevSize = count;
          prevToExpand.swap(toExpand);
          toExpand.clear();
          std::sort(prevToExpand.begin(), prevToExpand.end(),
                    [this](const std::pair<int, int>& a, const std::pair<int, int>& b) {
                      return a.first != b.first ? a.first < b.first : pool[a.second].seq < pool[b.second].seq;
                    });
          for (int j = (int)prevToExpand.size() - 1; j >= 0; j--) {
            int dummy = 0;
            divideAndPush(prevToExpand[j].second, x, y, dummy);
            erase(prevToExpand[j].second);
            if (count >= N) break;
          }
          if (count >= N || count == prevSize) finish = true;
        }
      }
    }

    out.reserve(count);
    for (int id = head; id >= 0; id = pool[id].next) {
      const NodeRec& nd = pool[id];
      const int* idx = ibuf[nd.buf].data();
      int best = idx[nd.begin];
      int maxResponse = score[best];
      for (int i = nd.begin + 1; i < nd.end; i++) {
        const int k = idx[i];
        if (score[k] > maxResponse) { best = k; maxResponse = score[k]; }
      }
      out.push_back(best);
    }
  }
};

}  // namespace orbfe
