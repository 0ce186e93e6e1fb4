"""Graph embeddings (reference: deeplearning4j-graph/**: graph/Graph.java, api/{Vertex,Edge,NoEdgeHandling}.java,
data/GraphLoader.java, iterator/{RandomWalkIterator,WeightedRandomWalkIterator}.java, models/deepwalk/{DeepWalk,GraphHuffman}.java,
models/loader/GraphVectorSerializer.java; and deeplearning4j-nlp models/node2vec/Node2Vec.java).

DeepWalk = truncated random walks + hierarchical-softmax skip-gram over vertex ids; training reuses the batched device
kernels of ``nlp.embeddings`` (whole walks are turned into pair batches, updated with gather/index_add)."""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from .nlp.embeddings import SequenceVectors
from .nlp.vocab import AbstractCache, Huffman, VocabWord


class NoEdgeHandling:
    SELF_LOOP_ON_DISCONNECTED, EXCEPTION_ON_DISCONNECTED = "SELF_LOOP_ON_DISCONNECTED", "EXCEPTION_ON_DISCONNECTED"


class NoEdgesException(RuntimeError):
    pass


class Vertex:
    def __init__(self, idx, value=None):
        self.idx, self.value = int(idx), value

    def vertexID(self):
        return self.idx

    def getValue(self):
        return self.value

    def __eq__(self, o):
        return isinstance(o, Vertex) and o.idx == self.idx and o.value == self.value

    def __hash__(self):
        return hash(self.idx)

    def __repr__(self):
        return "vertex(%d,%r)" % (self.idx, self.value)


class Edge:
    def __init__(self, frm, to, value=None, directed=False):
        self.frm, self.to, self.value, self.directed = int(frm), int(to), value, bool(directed)

    def getFrom(self):
        return self.frm

    def getTo(self):
        return self.to

    def getValue(self):
        return self.value

    def isDirected(self):
        return self.directed


class Graph:
    """Adjacency-list graph with integer vertex ids 0..n-1 (Graph.java)."""

    def __init__(self, vertices_or_n, allowMultipleEdges=False):
        if isinstance(vertices_or_n, int):
            self.vertices = [Vertex(i, i) for i in range(vertices_or_n)]
        else:
            self.vertices = list(vertices_or_n)
        self.allowMultipleEdges = allowMultipleEdges
        self.edges: List[List[Edge]] = [[] for _ in self.vertices]

    def numVertices(self):
        return len(self.vertices)

    def getVertex(self, i):
        return self.vertices[i]

    def getVertices(self, idx_or_from, to=None):
        if to is None:
            return [self.vertices[i] for i in idx_or_from]
        return self.vertices[idx_or_from:to + 1]

    def addEdge(self, frm, to=None, value=None, directed=False):
        e = frm if isinstance(frm, Edge) else Edge(frm, to, value, directed)
        if not (0 <= e.frm < len(self.vertices) and 0 <= e.to < len(self.vertices)):
            raise ValueError("Invalid edge: vertex index out of range")
        if not self.allowMultipleEdges and any((x.to == e.to and x.frm == e.frm) or (not x.directed and x.frm == e.to and x.to == e.frm)
                                               for x in self.edges[e.frm]):
            return
        self.edges[e.frm].append(e)
        if not e.directed and e.frm != e.to:
            self.edges[e.to].append(e)

    def getEdgesOut(self, v):
        return list(self.edges[v])

    def getVertexDegree(self, v):
        return len(self.edges[v])

    def _neighbors(self, v):
        return [(e.to if e.frm == v else e.frm) for e in self.edges[v] if (not e.directed or e.frm == v)]

    def getConnectedVertexIndices(self, v):
        return self._neighbors(v)

    def getConnectedVertices(self, v):
        return [self.vertices[i] for i in self._neighbors(v)]

    def getRandomConnectedVertex(self, v, rng):
        nb = self._neighbors(v)
        if not nb:
            raise NoEdgesException("Cannot generate random connected vertex: vertex %d has no outgoing/undirected edges" % v)
        return self.vertices[nb[int(rng.integers(0, len(nb)))]]

    def csr(self, weighted=False):
        """(indptr, indices, weights) arrays for vectorised walk generation."""
        indptr = np.zeros(len(self.vertices) + 1, dtype=np.int64)
        idx, w = [], []
        for v in range(len(self.vertices)):
            for e in self.edges[v]:
                if e.directed and e.frm != v:
                    continue
                idx.append(e.to if e.frm == v else e.frm)
                w.append(float(e.value) if (weighted and e.value is not None) else 1.0)
            indptr[v + 1] = len(idx)
        return indptr, np.asarray(idx, dtype=np.int64), np.asarray(w, dtype=np.float64)


class GraphLoader:
    @staticmethod
    def loadUndirectedGraphEdgeListFile(path, numVertices, delim=",", allowMultipleEdges=False) -> Graph:
        g = Graph(numVertices, allowMultipleEdges)
        with open(os.fspath(path), encoding="utf-8") as f:
            for line in f:
                line = line.strip()
                if not line or line.startswith("#") or line.startswith("//"):
                    continue
                a, b = line.split(delim)[:2]
                g.addEdge(int(a), int(b), None, False)
        return g

    @staticmethod
    def loadWeightedEdgeListFile(path, numVertices, delim=",", directed=False, allowMultipleEdges=False, ignoreLinesStartingWith=()) -> Graph:
        g = Graph(numVertices, allowMultipleEdges)
        with open(os.fspath(path), encoding="utf-8") as f:
            for line in f:
                line = line.strip()
                if not line or any(line.startswith(p) for p in ignoreLinesStartingWith):
                    continue
                a, b, w = line.split(delim)[:3]
                g.addEdge(int(a), int(b), float(w), directed)
        return g

    @staticmethod
    def loadGraph(vertexFile, edgeFile, delim=",", directed=False) -> Graph:
        with open(os.fspath(vertexFile), encoding="utf-8") as f:
            verts = [Vertex(i, l.strip()) for i, l in enumerate(f) if l.strip()]
        g = Graph(verts)
        with open(os.fspath(edgeFile), encoding="utf-8") as f:
            for line in f:
                p = line.strip().split(delim)
                if len(p) >= 2:
                    g.addEdge(int(p[0]), int(p[1]), float(p[2]) if len(p) > 2 else None, directed)
        return g


class RandomWalkIterator:
    """One uniform random walk of ``walkLength`` steps from every vertex, start order shuffled (RandomWalkIterator.java)."""
    WEIGHTED = False

    def __init__(self, graph: Graph, walkLength, seed=12345, mode=NoEdgeHandling.EXCEPTION_ON_DISCONNECTED, firstVertex=0, lastVertex=None):
        self.graph, self.walkLength, self.seed, self.mode = graph, int(walkLength), seed, mode
        self.first, self.last = firstVertex, graph.numVertices() if lastVertex is None else lastVertex
        self._epoch = 0
        self.reset()

    def walkLength_(self):
        return self.walkLength

    def reset(self):
        self.rng = np.random.default_rng(self.seed + self._epoch)
        self._epoch += 1
        self.order = self.first + self.rng.permutation(self.last - self.first)
        self.pos = 0

    def hasNext(self):
        return self.pos < len(self.order)

    def _step(self, cur, indptr, indices, weights):
        deg = indptr[cur + 1] - indptr[cur]
        if (deg == 0).any():
            if self.mode == NoEdgeHandling.EXCEPTION_ON_DISCONNECTED:
                raise NoEdgesException("Cannot conduct random walk: vertex %d has no outgoing edges" % int(cur[deg == 0][0]))
        r = self.rng.random(cur.shape[0])
        if self.WEIGHTED:
            nxt = cur.copy()
            for k in np.nonzero(deg > 0)[0]:
                lo, hi = indptr[cur[k]], indptr[cur[k] + 1]
                w = weights[lo:hi]
                nxt[k] = indices[lo + np.searchsorted(np.cumsum(w) / w.sum(), r[k])]
            return nxt
        off = np.minimum((r * np.maximum(deg, 1)).astype(np.int64), np.maximum(deg - 1, 0))
        return np.where(deg > 0, indices[np.minimum(indptr[cur] + off, max(len(indices) - 1, 0))] if len(indices) else cur, cur)

    def all_walks(self) -> np.ndarray:
        """Every remaining walk at once: int64 [n, walkLength + 1]."""
        indptr, indices, weights = self.graph.csr(self.WEIGHTED)
        cur = self.order[self.pos:].astype(np.int64)
        self.pos = len(self.order)
        out = np.empty((cur.shape[0], self.walkLength + 1), dtype=np.int64)
        out[:, 0] = cur
        for s in range(1, self.walkLength + 1):
            cur = self._step(cur, indptr, indices, weights)
            out[:, s] = cur
        return out

    def next(self) -> List[Vertex]:
        indptr, indices, weights = self.graph.csr(self.WEIGHTED)
        cur = np.array([self.order[self.pos]], dtype=np.int64)
        self.pos += 1
        walk = [int(cur[0])]
        for _ in range(self.walkLength):
            cur = self._step(cur, indptr, indices, weights)
            walk.append(int(cur[0]))
        return [self.graph.getVertex(i) for i in walk]

    def __iter__(self):
        while self.hasNext():
            yield self.next()


class WeightedRandomWalkIterator(RandomWalkIterator):
    """Next vertex chosen with probability proportional to edge weight."""
    WEIGHTED = True


class GraphHuffman:
    """Huffman tree over vertices keyed by degree (GraphHuffman.java): getCodeLength / getCode / getPathInnerNodes."""

    def __init__(self, nVertices):
        self.n = nVertices
        self._words: List[VocabWord] = []

    def buildTree(self, vertexDegree: Sequence[int]):
        self._words = [VocabWord(max(float(d), 1e-9), str(i), i) for i, d in enumerate(vertexDegree)]
        Huffman(self._words).build()

    def getCodeLength(self, v):
        return self._words[v].codeLength

    def getCode(self, v) -> int:
        c = 0
        for k, b in enumerate(self._words[v].codes):
            c |= b << k
        return c

    def getCodeList(self, v):
        return list(self._words[v].codes)

    def getPathInnerNodes(self, v):
        return list(self._words[v].points)


class DeepWalk:
    def __init__(self, vectorSize=100, windowSize=2, learningRate=0.01, seed=12345, batchSize=8192, device=None):
        self.vectorSize, self.windowSize, self.learningRate, self.seed = int(vectorSize), int(windowSize), float(learningRate), seed
        self.batchSize, self.device = batchSize, device
        self._sv: Optional[SequenceVectors] = None
        self._n = 0
        self._init = False

    class Builder:
        def __init__(self):
            self._kw = {}

        def vectorSize(self, n):
            self._kw["vectorSize"] = n
            return self

        def windowSize(self, n):
            self._kw["windowSize"] = n
            return self

        def learningRate(self, lr):
            self._kw["learningRate"] = lr
            return self

        def seed(self, s):
            self._kw["seed"] = s
            return self

        def build(self):
            return DeepWalk(**self._kw)

    def initialize(self, graph_or_degrees):
        deg = [graph_or_degrees.getVertexDegree(i) for i in range(graph_or_degrees.numVertices())] if isinstance(graph_or_degrees, Graph) \
            else list(graph_or_degrees)
        self._n = len(deg)
        vocab = AbstractCache()
        for i, d in enumerate(deg):
            vocab.addToken(VocabWord(max(float(d), 1.0), str(i), i))
            vocab.addWordToIndex(i, str(i))                      # vertex id == table row (no frequency re-ordering)
        Huffman(vocab.vocabWords()).build()
        self._sv = SequenceVectors(minWordFrequency=1, layerSize=self.vectorSize, windowSize=self.windowSize, learningRate=self.learningRate,
                                   minLearningRate=self.learningRate, useHierarchicSoftmax=True, negativeSample=0, seed=self.seed,
                                   batchSize=self.batchSize, device=self.device, vocabCache=vocab, resetModel=False, sequences=[])
        self._init = True

    def fit(self, graph_or_iterator, walkLength=None):
        if isinstance(graph_or_iterator, Graph):
            if not self._init:
                self.initialize(graph_or_iterator)
            it = RandomWalkIterator(graph_or_iterator, walkLength, self.seed)
        else:
            it = graph_or_iterator
            if not self._init:
                raise RuntimeError("DeepWalk not initialized (call initialize before fit)")
        walks = it.all_walks() if hasattr(it, "all_walks") else np.array([[v.vertexID() for v in w] for w in it], dtype=np.int64)
        self._sv._sequences = [[str(int(v)) for v in row] for row in walks]
        first = self._sv.lookupTable is None
        self._sv.resetModel = first
        self._sv.fit()
        self._sv.resetModel = False
        return self

    # -- GraphVectors API
    def numVertices(self):
        return self._n

    def getVectorSize(self):
        return self.vectorSize

    def getVertexVector(self, v):
        v = v.vertexID() if isinstance(v, Vertex) else int(v)
        return self._sv.lookupTable.syn0[v]

    def getVertexVectors(self):
        return self._sv.lookupTable.syn0

    def similarity(self, a, b):
        a = a.vertexID() if isinstance(a, Vertex) else int(a)
        b = b.vertexID() if isinstance(b, Vertex) else int(b)
        if a == b:
            return 1.0
        va, vb = self.getVertexVector(a), self.getVertexVector(b)
        return float(torch.dot(va, vb) / (va.norm() * vb.norm()).clamp_min(1e-12))

    def verticesNearest(self, v, top):
        v = v.vertexID() if isinstance(v, Vertex) else int(v)
        m = self._sv.lookupTable.syn0
        u = m / m.norm(dim=1, keepdim=True).clamp_min(1e-12)
        sims = u @ u[v]
        sims[v] = -2
        return torch.topk(sims, min(top, self._n - 1)).indices.tolist()


class GraphVectorSerializer:
    @staticmethod
    def writeGraphVectors(dw: DeepWalk, path):
        m = dw.getVertexVectors().detach().cpu().double().numpy()
        with open(os.fspath(path), "w", encoding="utf-8") as f:
            for i, row in enumerate(m):
                f.write(str(i) + "\t" + "\t".join(repr(float(x)) for x in row) + "\n")

    @staticmethod
    def loadTxtVectors(path, device=None) -> DeepWalk:
        rows = []
        with open(os.fspath(path), encoding="utf-8") as f:
            for line in f:
                p = line.rstrip("\n").split("\t")
                if len(p) > 1:
                    rows.append([float(x) for x in p[1:]])
        m = np.asarray(rows, dtype=np.float32)
        dw = DeepWalk(vectorSize=m.shape[1], device=device)
        dw.initialize([1] * m.shape[0])
        from .nlp.embeddings import InMemoryLookupTable
        dw._sv.lookupTable = InMemoryLookupTable(dw._sv.vocab, m.shape[1], True, 0, device=device)
        dw._sv.lookupTable.syn0 = torch.from_numpy(m).to(dw._sv.lookupTable.device)
        return dw


class Node2Vec(SequenceVectors):
    """node2vec: second-order biased walks (return parameter p, in-out parameter q) + skip-gram (Node2Vec.java)."""

    def __init__(self, graph: Graph = None, walkLength=40, walksPerVertex=10, p=1.0, q=1.0, **kw):
        self.graph, self.walkLength, self.walksPerVertex, self.p, self.q = graph, int(walkLength), int(walksPerVertex), float(p), float(q)
        kw.setdefault("minWordFrequency", 1)
        kw.setdefault("useHierarchicSoftmax", False)
        kw.setdefault("negativeSample", 5)
        super().__init__(**kw)

    def _walks(self):
        g, rng = self.graph, np.random.default_rng(self.seed)
        nbrs = [g.getConnectedVertexIndices(v) for v in range(g.numVertices())]
        nset = [set(n) for n in nbrs]
        walks = []
        for _ in range(self.walksPerVertex):
            for start in rng.permutation(g.numVertices()):
                walk = [int(start)]
                while len(walk) < self.walkLength + 1:
                    cur = walk[-1]
                    nb = nbrs[cur]
                    if not nb:
                        break
                    if len(walk) == 1:
                        walk.append(nb[int(rng.integers(0, len(nb)))])
                        continue
                    prev = walk[-2]
                    w = np.array([1.0 / self.p if x == prev else (1.0 if x in nset[prev] else 1.0 / self.q) for x in nb])
                    walk.append(nb[int(np.searchsorted(np.cumsum(w) / w.sum(), rng.random()))])
                walks.append([str(v) for v in walk])
        return walks

    def fit(self):
        self._sequences = self._walks()
        return super().fit()

    def getVertexVector(self, v):
        return self.lookupTable.vector(str(int(v)))

    def verticesNearest(self, v, top=10):
        return [int(w) for w in self.wordsNearest(str(int(v)), top)]
