#!/usr/bin/env python3
"""Lab input: the index lists of one workload in several internal orders, as raw arrays for scripts/lab/spmm_lab.hip.
usage: python scripts/lab/make_plan.py <workload> <outdir> [cluster size]
Writes <outdir>/<variant>/{meta.txt, starts.u32, pairs.u32, chunkFirst.u32, chunkCol.u32, order.u32, rowI.u32, origCol.i32}
variants: base (the product's order: column, row; chunks of 4 blocks), c16 (rows clustered into patches, chunk = column x cluster)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tfqmrgpu_amd as T
from bench import build_problem


def clusters_of(adj, size):
    """greedy patches: grow a cluster from a seed by adding the unvisited neighbour with most links into the cluster,
    ties to the one closest to the seed (hops inside the cluster), so that patches stay round; the next seed is the
    unvisited row with most links to finished clusters (keeps fragments small)"""
    n = len(adj)
    cl = -np.ones(n, np.int64)
    order = []
    done_links = np.zeros(n, np.int64)     # links of an unvisited row into finished clusters
    ncl = 0
    import heapq
    heap = [(0, r) for r in range(n)]      # (-done_links, row), lazily updated
    heapq.heapify(heap)
    while heap:
        negl, s = heapq.heappop(heap)
        if cl[s] >= 0 or -negl != done_links[s]:
            continue
        members = [s]; cl[s] = ncl
        dist = {s: 0}
        links = {}
        def touch(r):
            for q in adj[r]:
                if cl[q] < 0:
                    links[q] = links.get(q, 0) + 1
                    dist[q] = min(dist.get(q, 1 << 30), dist[r] + 1)
        touch(s)
        while len(members) < size and links:
            best = max(links.items(), key=lambda kv: (kv[1], -dist[kv[0]], done_links[kv[0]], -kv[0]))[0]
            del links[best]
            cl[best] = ncl; members.append(best); touch(best)
        order += sorted(members)
        for m in members:
            for q in adj[m]:
                if cl[q] < 0:
                    done_links[q] += 1
                    heapq.heappush(heap, (-done_links[q], q))
        ncl += 1
    return cl, np.array(order)


def main():
    name, out = sys.argv[1], sys.argv[2]
    csize = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    pr, prec, desc = build_problem(name, 0)
    with T.Solver() as s:
        s.create_plan(pr)
        v = s.plan_view()
    nX, nA, nP, nCols = pr.nnzbX, pr.nnzbA, v["nPairs"], v["nCols"]
    rowOfX = np.repeat(np.arange(pr.mb), np.diff(pr.rowPtrX))
    col = v["colindx"].astype(np.int64)
    starts, pairs = v["starts"].astype(np.int64), v["pairs"].astype(np.int64).reshape(-1, 2)
    rowsA = np.repeat(np.arange(pr.mb), np.diff(pr.rowPtrA))
    adj = [[] for _ in range(pr.mb)]
    for r, c in zip(rowsA, pr.colIndA - pr.index_offset):
        if r != c:
            adj[r].append(int(c)); adj[int(c)].append(int(r))
    adj = [sorted(set(a)) for a in adj]

    def emit(variant, rank, cluster, chunk_blocks, G=4):
        """internal order: (column, rank[row]); chunks: runs inside one (column, cluster) of at most chunk_blocks blocks"""
        i2u = np.lexsort((rank[rowOfX], col))
        u2i = np.empty(nX, np.int64); u2i[i2u] = np.arange(nX)
        cnt = (starts[1:] - starts[:-1])[i2u]
        st = np.concatenate([[0], np.cumsum(cnt)])
        src = np.concatenate([np.arange(starts[u], starts[u + 1]) for u in i2u]) if nP else np.zeros(0, np.int64)
        pa = np.stack([pairs[src, 0], u2i[pairs[src, 1]]], axis=1)
        ci, ri = col[i2u], rowOfX[i2u]
        key = ci * (cluster.max() + 1) + cluster[ri]
        first, ccol = [], []
        b = 0
        while b < nX:
            e = b
            while e < nX and key[e] == key[b] and e - b < chunk_blocks:
                e += 1
            first.append(b); ccol.append(ci[b]); b = e
        first.append(nX)
        first, ccol = np.array(first), np.array(ccol)
        n = len(ccol)
        # launch order like tfq_plan.cpp: (group of 4 columns, band, column), 8 contiguous parts dealt round-robin
        band = rank[ri[first[:-1]]] // max(1, chunk_blocks)
        so = np.lexsort((ccol, band, ccol // G))
        q, r = divmod(n, 8)
        begin = np.concatenate([[0], np.cumsum([q + (1 if x < r else 0) for x in range(8)])])
        order = []
        for i in range(q + 1):
            for x in range(8):
                if begin[x] + i < begin[x + 1]:
                    order.append(so[begin[x] + i])
        d = os.path.join(out, variant); os.makedirs(d, exist_ok=True)
        # in-chunk pairs first inside every Y block (stable), split point per Y block
        chunk_of = np.repeat(np.arange(n), np.diff(first))
        lo, hi = first[chunk_of], first[chunk_of + 1]
        yb = np.repeat(np.arange(nX), cnt)
        inside = (pa[:, 1] >= lo[yb]) & (pa[:, 1] < hi[yb])
        o2 = np.lexsort((~inside, yb))
        pa2 = pa[o2]
        nin = np.bincount(yb, weights=inside, minlength=nX).astype(np.int64)
        for nm, arr, dt in (("starts", st, np.uint32), ("pairs", pa2.reshape(-1), np.uint32), ("nInside", nin, np.uint32),
                            ("chunkFirst", first, np.uint32), ("chunkCol", ccol, np.uint32), ("order", np.array(order), np.uint32),
                            ("rowI", ri, np.uint32), ("origCol", v["original_bsrColIndX"] - pr.index_offset, np.int32)):
            np.asarray(arr).astype(dt).tofile(os.path.join(d, nm + ".bin"))
        with open(os.path.join(d, "meta.txt"), "w") as f:
            f.write("%d %d %d %d %d %d %d\n" % (nX, nA, nP, n, nCols, pr.LM, pr.LN))
        uniq_halo = 0
        print("%-8s chunks %6d (max %d blocks, mean %.2f), pairs inside their chunk %.1f %%" % (
            variant, n, np.diff(first).max(), np.diff(first).mean(), 100.0 * inside.mean()))

    ident = np.arange(pr.mb)
    emit("base", ident, ident * 0, 4)
    if os.environ.get("LAB_BASE_ONLY"):
        return
         # cluster = row: chunks cut every 4 blocks (the product's 16 KiB chunks for 16x16 z)
    emit("line16", ident, ident // csize, csize)
    cl, order = clusters_of(adj, csize)
    rank = np.empty(pr.mb, np.int64); rank[order] = np.arange(pr.mb)
    emit("c%d" % csize, rank, cl, csize)
    for G in (16, 49):
        emit("c%dg%d" % (csize, G), rank, cl, csize, G)


if __name__ == "__main__":
    main()
