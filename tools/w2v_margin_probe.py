import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_nlp_cpu import _corpus, _topic_margin
from deeplearning4j_amd.nlp import Word2Vec
sents, a, b = _corpus()
for algo, hs, neg in (("CBOW", True, 0), ("SkipGram", True, 3), ("SkipGram", False, 5)):
    for ep in (25, 40):
        ms = []
        for rep in range(4):
            m = (Word2Vec.Builder().minWordFrequency(1).layerSize(24).windowSize(4).epochs(ep).learningRate(0.05).seed(1).iterate(sents)
                 .useHierarchicSoftmax(hs).negativeSample(neg).elementsLearningAlgorithm(algo).batchSize(512).device("cuda").build())
            m.fit()
            ms.append(round(float(_topic_margin(m, a, b)), 3))
        print(algo, hs, neg, "epochs", ep, ms, flush=True)
