"""NLP (reference: deeplearning4j-nlp-parent/deeplearning4j-nlp)."""
from .tokenization import (TokenPreProcess, CommonPreprocessor, LowCasePreProcessor, EndingPreProcessor, CompositePreProcessor,  # noqa: F401
                           BertWordPiecePreProcessor, Tokenizer, TokenizerFactory, DefaultTokenizerFactory, NGramTokenizerFactory,
                           BertWordPieceTokenizer, BertWordPieceTokenizerFactory, StopWords)
from .iterators import (SentenceIterator, CollectionSentenceIterator, BasicLineIterator, LineSentenceIterator, FileSentenceIterator,  # noqa: F401
                        AggregatingSentenceIterator, LabelledDocument, LabelsSource, LabelAwareIterator, SimpleLabelAwareIterator,
                        BasicLabelAwareIterator, FileLabelAwareIterator, LabelAwareListSentenceIterator)
from .vocab import VocabWord, AbstractCache, InMemoryLookupCache, Huffman, VocabConstructor  # noqa: F401
from .embeddings import InMemoryLookupTable, WordVectorsImpl, SequenceVectors, Word2Vec, ParagraphVectors  # noqa: F401
from .fasttext import FastText  # noqa: F401
from .serializer import WordVectorSerializer  # noqa: F401
from .bert import BertIterator, BertMaskedLMMasker, CollectionLabeledSentenceProvider, CollectionLabeledPairSentenceProvider  # noqa: F401
from .text import (BagOfWordsVectorizer, TfidfVectorizer, CnnSentenceDataSetIterator, UnknownWordHandling, Format, Window, Windows,  # noqa: F401,E402
                   WindowConverter, ContextLabelRetriever, InvertedIndex, StaticWord2Vec)
