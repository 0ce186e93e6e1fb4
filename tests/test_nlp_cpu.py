"""NLP stack: tokenizers, vocab/Huffman, Word2Vec (skip-gram / CBOW, HS / NS), ParagraphVectors, FastText, serializer, BertIterator,
DeepWalk / Node2Vec.  Reference tests: Word2VecTests, ParagraphVectorsTest, WordVectorSerializerTest, BertWordPieceTokenizerTests,
TestBertIterator, DeepWalkGradientCheck / TestDeepWalk, FastTextTest."""
import numpy as np
import pytest
import torch

from deeplearning4j_amd.graph_embeddings import (DeepWalk, Graph, GraphHuffman, GraphLoader, GraphVectorSerializer, Node2Vec,
                                                 NoEdgesException, RandomWalkIterator, WeightedRandomWalkIterator)
from deeplearning4j_amd.nlp import (AbstractCache, BasicLineIterator, BertIterator, BertWordPieceTokenizerFactory,
                                    CollectionLabeledPairSentenceProvider, CollectionLabeledSentenceProvider, CollectionSentenceIterator,
                                    CommonPreprocessor, DefaultTokenizerFactory, FastText, Huffman, LabelledDocument, NGramTokenizerFactory,
                                    ParagraphVectors, VocabConstructor, VocabWord, Word2Vec, WordVectorSerializer)
from deeplearning4j_amd.nlp.embeddings import InMemoryLookupTable, _hs_tables, _train_batch


def _corpus(n=300, seed=0):
    """Two topics with disjoint vocabularies: words of one topic co-occur, so they must end up closer than cross-topic words."""
    rng = np.random.default_rng(seed)
    a = ["cat", "dog", "horse", "cow", "sheep", "goat"]
    b = ["car", "bus", "train", "plane", "boat", "truck"]
    out = []
    for i in range(n):
        topic = a if i % 2 == 0 else b
        out.append(" ".join(rng.choice(topic, 8)))
    return out, a, b


def _topic_margin(m, a, b):
    same = np.mean([m.similarity(x, y) for x in a for y in a if x != y] + [m.similarity(x, y) for x in b for y in b if x != y])
    cross = np.mean([m.similarity(x, y) for x in a for y in b])
    return same - cross


# ----------------------------------------------------------------------------- tokenizers
def test_tokenizers_and_preprocessors():
    tf = DefaultTokenizerFactory().setTokenPreProcessor(CommonPreprocessor())
    assert tf.create("Hello, World! It's 2024.").getTokens() == ["hello", "world", "its"]
    ng = NGramTokenizerFactory(DefaultTokenizerFactory(), 1, 2)
    assert ng.create("a b c").getTokens() == ["a", "b", "c", "a b", "b c"]
    t = DefaultTokenizerFactory().create("x y")
    assert t.countTokens() == 2 and t.nextToken() == "x" and t.hasMoreTokens() and t.nextToken() == "y" and not t.hasMoreTokens()


def test_bert_wordpiece():
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "un", "##aff", "##able", "i", "like", "playing", "play", "##ing", ",", "cafe"]
    tf = BertWordPieceTokenizerFactory(vocab, lowerCaseOnly=True, stripAccents=True)
    assert tf.create("unaffable").getTokens() == ["un", "##aff", "##able"]
    assert tf.create("I like PLAYING, xyzzy").getTokens() == ["i", "like", "playing", ",", "[UNK]"]
    assert tf.create("Café").getTokens() == ["cafe"]


# ----------------------------------------------------------------------------- vocab / huffman
def test_huffman_codes_are_prefix_free_and_optimal():
    freqs = [50, 30, 10, 5, 3, 2]
    ws = [VocabWord(f, "w%d" % i, i) for i, f in enumerate(freqs)]
    Huffman(ws).build()
    codes = ["".join(map(str, w.codes)) for w in ws]
    assert all(len(c) == w.codeLength and len(w.points) == w.codeLength for c, w in zip(codes, ws))
    for i, c in enumerate(codes):
        for j, d in enumerate(codes):
            assert i == j or not d.startswith(c)
    assert sum(2.0 ** -len(c) for c in codes) == 1.0                       # full binary tree (Kraft equality)
    assert [len(c) for c in codes] == sorted(len(c) for c in codes)        # frequent words get short codes
    assert all(w.points[0] == len(ws) - 2 for w in ws)                     # root first
    assert all(0 <= p < len(ws) - 1 for w in ws for p in w.points)


def test_vocab_constructor_min_frequency_and_order():
    seqs = [["a", "b", "a", "c"], ["a", "b", "d"]]
    v = VocabConstructor(minWordFrequency=2).buildJointVocabulary(seqs)
    assert v.words() == ["a", "b"] and v.wordFrequency("a") == 3 and v.indexOf("c") == -1
    assert v.totalNumberOfDocs() == 2 and v.docAppearedIn("a") == 2 and v.totalWordOccurrences() == 5


# ----------------------------------------------------------------------------- learning kernels
@pytest.mark.parametrize("use_hs,neg", [(True, 0), (False, 4)])
def test_batched_step_matches_per_sample_reference(use_hs, neg):
    """One batch with distinct rows == the textbook per-sample word2vec update (sg_cb.cu semantics) applied independently."""
    torch.manual_seed(0)
    ws = [VocabWord(10 - i, "w%d" % i, i) for i in range(8)]
    cache = AbstractCache()
    for w in ws:
        cache.addToken(w)
    cache.rebuild_index()
    Huffman(cache.vocabWords()).build()
    t = InMemoryLookupTable(cache, 6, use_hs, neg, device="cpu")
    t.resetWeights()
    if t.syn1 is not None:
        t.syn1.normal_(0, 0.3)
    if t.syn1neg is not None:
        t.syn1neg.normal_(0, 0.3)
    inp, tg = torch.tensor([0, 1]), torch.tensor([2, 3])
    hs = _hs_tables(cache, "cpu") if use_hs else None
    syn1_0 = None if t.syn1 is None else t.syn1.clone()
    l1 = t.syn0[inp].clone()
    gen = torch.Generator().manual_seed(5)
    neu1e = _train_batch(t, l1, tg, 0.05, hs, 0 if use_hs else neg, gen)
    if use_hs:
        for b in range(2):
            w = cache.elementAtIndex(int(tg[b]))
            e = torch.zeros(6)
            for c, p in zip(w.codes, w.points):
                f = torch.sigmoid(syn1_0[p] @ l1[b])
                g = (1 - c - f) * 0.05
                e += g * syn1_0[p]
            assert torch.allclose(neu1e[b], e, atol=1e-6)
        # rows of syn1 touched by exactly one sample moved by g * l1
        w = cache.elementAtIndex(2)
        other = set(cache.elementAtIndex(3).points)
        for c, p in zip(w.codes, w.points):
            if p not in other:
                g = (1 - c - torch.sigmoid(syn1_0[p] @ l1[0])) * 0.05
                assert torch.allclose(t.syn1[p], syn1_0[p] + g * l1[0], atol=1e-6)
    else:
        assert neu1e.shape == (2, 6) and torch.isfinite(neu1e).all() and float(neu1e.abs().sum()) > 0


# ----------------------------------------------------------------------------- Word2Vec
@pytest.mark.parametrize("algo,hs,neg", [("SkipGram", True, 0), ("SkipGram", False, 5), ("CBOW", True, 0), ("CBOW", False, 5)])
def test_word2vec_learns_topics(algo, hs, neg):
    sents, a, b = _corpus()
    m = (Word2Vec.Builder().minWordFrequency(1).layerSize(24).windowSize(4).epochs(25).learningRate(0.05).seed(1).iterate(sents)
         .useHierarchicSoftmax(hs).negativeSample(neg).elementsLearningAlgorithm(algo).batchSize(512).device("cpu").build())
    m.fit()
    assert m.vocab.numWords() == 12
    assert _topic_margin(m, a, b) > 0.25
    near = m.wordsNearest("cat", 5)
    assert len(near) == 5 and "cat" not in near and sum(w in a for w in near) >= 4
    assert m.hasWord("dog") and not m.hasWord("zebra") and m.getWordVector("dog").shape == (24,)
    assert abs(m.similarity("cat", "cat") - 1.0) < 1e-9


def test_word2vec_file_iterator_and_serializer_roundtrip(tmp_path):
    sents, a, b = _corpus(120)
    p = tmp_path / "corpus.txt"
    p.write_text("\n".join(sents) + "\n")
    m = Word2Vec(minWordFrequency=1, layerSize=16, windowSize=3, epochs=3, seed=3, iterate=BasicLineIterator(p), negativeSample=3, device="cpu")
    m.fit()
    z = tmp_path / "w2v.zip"
    WordVectorSerializer.writeWord2VecModel(m, z)
    import zipfile
    assert set(zipfile.ZipFile(z).namelist()) >= {"syn0.txt", "syn1.txt", "syn1Neg.txt", "codes.txt", "huffman.txt", "frequencies.txt", "config.json"}
    first = zipfile.ZipFile(z).read("syn0.txt").decode().splitlines()
    assert first[0] == "12 16 120" and first[1].startswith("B64:")
    r = WordVectorSerializer.readWord2VecModel(z, device="cpu")
    assert r.vocab.words() == m.vocab.words()
    assert torch.allclose(r.lookupTable.syn0, m.lookupTable.syn0, atol=1e-6)
    assert torch.allclose(r.lookupTable.syn1, m.lookupTable.syn1, atol=1e-6) and torch.allclose(r.lookupTable.syn1neg, m.lookupTable.syn1neg, atol=1e-6)
    w = m.vocab.wordFor("cat")
    rw = r.vocab.wordFor("cat")
    assert rw.codes == w.codes and rw.points == w.points and rw.elementFrequency == w.elementFrequency
    # continue training a restored model
    r.sentenceIterator = sents
    r.fit()
    # text + google binary formats
    WordVectorSerializer.writeWordVectors(m, tmp_path / "v.txt")
    t = WordVectorSerializer.readWord2VecModel(tmp_path / "v.txt", device="cpu")
    assert t.wordsNearest("cat", 3) == m.wordsNearest("cat", 3)
    WordVectorSerializer.writeGoogleBinary(m, tmp_path / "v.bin")
    g = WordVectorSerializer.readWord2VecModel(tmp_path / "v.bin", device="cpu")
    assert g.vocab.words() == m.vocab.words() and torch.allclose(g.lookupTable.syn0, m.lookupTable.syn0)


# ----------------------------------------------------------------------------- ParagraphVectors
@pytest.mark.parametrize("algo", ["DBOW", "DM"])
def test_paragraph_vectors_classify(algo, tmp_path):
    sents, a, b = _corpus(200, seed=2)
    docs = [LabelledDocument(s, ["animals" if i % 2 == 0 else "vehicles"]) for i, s in enumerate(sents)]
    pv = (ParagraphVectors.Builder().iterate(docs).layerSize(20).windowSize(3).epochs(6).learningRate(0.05).negativeSample(5)
          .useHierarchicSoftmax(False).sequenceLearningAlgorithm(algo).seed(7).device("cpu").build())
    pv.fit()
    assert sorted(pv.getLabelsSource().getLabels()) == ["animals", "vehicles"]
    assert pv.vocab.wordFor("animals").isLabel and "animals" not in pv.wordsNearest("cat", 11)
    assert pv.predict("dog cow sheep goat horse cat") == "animals"
    assert pv.predict("bus train plane truck boat car") == "vehicles"
    assert pv.similarityToLabel("dog cow sheep", "animals") > pv.similarityToLabel("dog cow sheep", "vehicles")
    WordVectorSerializer.writeParagraphVectors(pv, tmp_path / "pv.zip")
    r = WordVectorSerializer.readParagraphVectors(tmp_path / "pv.zip", device="cpu")
    assert r.predict("bus train plane truck") == "vehicles"


# ----------------------------------------------------------------------------- FastText
def test_fasttext_unsupervised_oov_and_supervised(tmp_path):
    sents, a, b = _corpus(200, seed=4)
    ft = FastText(skipgram=True, iterator=sents, dim=16, epochs=20, batchSize=256, learningRate=0.1, bucket=5000, minNgramLength=2, maxNgramLength=4, negativeSamples=4,
                  samplingThreshold=0, device="cpu", seed=1)
    ft.fit()
    assert _topic_margin(ft, a, b) > 0.2
    oov = ft.getWordVector("horses")                      # unseen word: built from its character n-grams
    assert oov is not None and oov.shape == (16,)
    v = torch.tensor(oov, dtype=torch.float32)
    assert "horse" in ft.wordsNearest(v, 3)
    sup = ["__label__animal " + s if i % 2 == 0 else "__label__vehicle " + s for i, s in enumerate(sents)]
    f = tmp_path / "train.txt"
    f.write_text("\n".join(sup))
    clf = FastText.builder().supervised().inputFile(f).dim(10).epochs(10).wordNgrams(2).bucket(2000).learningRate(0.5).device("cpu").build()
    clf.fit()
    assert clf.getLabels() == ["__label__animal", "__label__vehicle"] or set(clf.getLabels()) == {"__label__animal", "__label__vehicle"}
    assert clf.predict("cow goat sheep") == "__label__animal" and clf.predict("truck bus car") == "__label__vehicle"
    lab, prob = clf.predictProbability("plane boat train")
    assert lab == "__label__vehicle" and 0.5 < prob <= 1.0
    clf.saveModel(tmp_path / "ft.bin")
    again = FastText.loadModel(tmp_path / "ft.bin", device="cpu")
    assert again.predict("cow goat sheep") == "__label__animal"


# ----------------------------------------------------------------------------- BertIterator
def _bert_tf():
    vocab = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "the", "cat", "sat", "on", "mat", "dog", "ran", "fast", "##s"]
    return BertWordPieceTokenizerFactory(vocab, lowerCaseOnly=True)


def test_bert_iterator_classification_and_pairs():
    tf = _bert_tf()
    prov = CollectionLabeledSentenceProvider(["the cat sat", "the dog ran fast", "cats"], ["pos", "neg", "pos"])
    it = (BertIterator.builder().tokenizer(tf).lengthHandling(BertIterator.LengthHandling.FIXED_LENGTH, 6).minibatchSize(2)
          .sentenceProvider(prov).featureArrays(BertIterator.FeatureArrays.INDICES_MASK_SEGMENTID).task(BertIterator.Task.SEQ_CLASSIFICATION)
          .appendToken("[SEP]").build())
    b1 = it.next()
    assert b1.features[0].tolist() == [[2, 5, 6, 7, 3, 0], [2, 5, 10, 11, 12, 3]]
    assert b1.featuresMasks[0].tolist() == [[1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1]]
    assert b1.labels[0].tolist() == [[0, 1], [1, 0]]                       # label order: sorted -> [neg, pos]
    b2 = it.next()
    assert b2.features[0].shape == (1, 6) and b2.features[0][0, :4].tolist() == [2, 6, 13, 3] and not it.hasNext()
    pp = CollectionLabeledPairSentenceProvider(["the cat", "the dog ran fast on the mat"], ["sat", "the cat sat on the mat"], ["a", "b"])
    itp = BertIterator(tf, sentencePairProvider=pp, maxTokens=8, minibatchSize=2, featureArrays="INDICES_MASK_SEGMENTID", appendToken="[SEP]")
    b = itp.next()
    assert b.features[0][0].tolist() == [2, 5, 6, 3, 7, 3, 0, 0] and b.features[1][0].tolist() == [0, 0, 0, 0, 1, 1, 0, 0]
    assert int(b.featuresMasks[0][1].sum()) == 8 and b.features[0][1, 0] == 2 and 3 in b.features[0][1].tolist()


def test_bert_iterator_masked_lm():
    tf = _bert_tf()
    prov = CollectionLabeledSentenceProvider(["the cat sat on the mat"] * 8)
    it = BertIterator(tf, task="UNSUPERVISED", lengthHandling="ANY_LENGTH", minibatchSize=8, sentenceProvider=prov)
    b = it.next()
    idx, lm, lab = b.features[0], b.labelsMasks[0], b.labels[0]
    assert idx.shape == (8, 7) and lab.shape == (8, 7)
    assert lab[0].tolist() == [2, 5, 6, 7, 8, 5, 9]                        # labels carry the *original* ids
    assert 0 < float(lm.sum()) < lm.numel()
    changed = idx != lab
    assert bool((lm[changed] == 1).all())                                 # only masked positions may differ
    it3 = BertIterator(tf, task="UNSUPERVISED", lengthHandling="ANY_LENGTH", minibatchSize=8, sentenceProvider=prov,
                       unsupervisedLabelFormat="RANK3_NCL")
    it3.reset()                                                           # the provider is shared with ``it`` and already consumed
    assert it3.next().labels[0].shape == (8, 14, 7)


# ----------------------------------------------------------------------------- graphs
def _two_cliques(n=6):
    g = Graph(2 * n)
    for base in (0, n):
        for i in range(n):
            for j in range(i + 1, n):
                g.addEdge(base + i, base + j)
    g.addEdge(0, n)                                                       # one bridge
    return g


def test_graph_api_walks_and_huffman(tmp_path):
    g = _two_cliques(4)
    assert g.numVertices() == 8 and g.getVertexDegree(0) == 4 and g.getVertexDegree(1) == 3
    assert sorted(g.getConnectedVertexIndices(0)) == [1, 2, 3, 4]
    it = RandomWalkIterator(g, 5, seed=1)
    walks = [w for w in it]
    assert len(walks) == 8 and sorted(w[0].vertexID() for w in walks) == list(range(8))
    for w in walks:
        assert len(w) == 6
        for x, y in zip(w, w[1:]):
            assert y.vertexID() in g.getConnectedVertexIndices(x.vertexID())
    lonely = Graph(3)
    lonely.addEdge(0, 1)
    with pytest.raises(NoEdgesException):
        RandomWalkIterator(lonely, 2).all_walks()
    f = tmp_path / "edges.csv"
    f.write_text("0,1,1.0\n1,2,100.0\n")
    wg = GraphLoader.loadWeightedEdgeListFile(f, 3)
    ww = WeightedRandomWalkIterator(wg, 1, seed=0)
    nxt = [w[1].vertexID() for w in ww if w[0].vertexID() == 1]
    assert nxt == [2]                                                     # weight 100 vs 1
    gh = GraphHuffman(8)
    gh.buildTree([g.getVertexDegree(i) for i in range(8)])
    assert gh.getCodeLength(0) <= gh.getCodeLength(1) and len(gh.getPathInnerNodes(3)) == gh.getCodeLength(3)


def test_deepwalk_and_node2vec_separate_communities(tmp_path):
    g = _two_cliques(6)
    dw = DeepWalk.Builder().vectorSize(16).windowSize(3).learningRate(0.05).seed(3).build()
    dw.device = "cpu"
    dw.initialize(g)
    for _ in range(30):
        dw.fit(g, 10)
    inside = np.mean([dw.similarity(1, j) for j in range(2, 6)])
    across = np.mean([dw.similarity(1, j) for j in range(7, 12)])
    assert inside > across + 0.2
    assert all(v < 6 for v in dw.verticesNearest(2, 3))
    GraphVectorSerializer.writeGraphVectors(dw, tmp_path / "gv.txt")
    back = GraphVectorSerializer.loadTxtVectors(tmp_path / "gv.txt", device="cpu")
    assert torch.allclose(back.getVertexVectors(), dw.getVertexVectors().float(), atol=1e-6)
    n2v = Node2Vec(g, walkLength=10, walksPerVertex=20, p=1.0, q=0.5, layerSize=16, windowSize=3, epochs=3, learningRate=0.05, device="cpu", seed=2)
    n2v.fit()
    assert sum(v >= 6 for v in n2v.verticesNearest(8, 4)) >= 3
