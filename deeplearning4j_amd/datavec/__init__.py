"""DataVec: ETL (reference: datavec/datavec-api, datavec-data-image, datavec-local)."""
from .writable import (Writable, IntWritable, LongWritable, FloatWritable, DoubleWritable, BooleanWritable, Text, NullWritable,  # noqa: F401
                       BytesWritable, NDArrayWritable)
from .records import (InputSplit, FileSplit, CollectionInputSplit, NumberedFileInputSplit, StringSplit, ListStringSplit, RecordReader,  # noqa: F401
                      SequenceRecordReader, LineRecordReader, CSVRecordReader, CSVSequenceRecordReader, CSVRegexRecordReader,
                      RegexLineRecordReader, SVMLightRecordReader, LibSvmRecordReader, JacksonLineRecordReader, CollectionRecordReader,
                      CollectionSequenceRecordReader, InMemoryRecordReader, InMemorySequenceRecordReader, FileRecordReader,
                      TransformProcessRecordReader, ConcatenatingRecordReader, Record, RecordMetaData)
from .transform import (ColumnType, Schema, TransformProcess, LocalTransformExecutor, AnalyzeLocal, DataAnalysis, ConditionOp,  # noqa: F401
                        ColumnCondition, DoubleColumnCondition, IntegerColumnCondition, LongColumnCondition, StringColumnCondition,
                        CategoricalColumnCondition, NullWritableColumnCondition, NaNColumnCondition, InvalidValueColumnCondition,
                        BooleanCondition, ConditionFilter, FilterInvalidValues, MathOp, MathFunction, ReduceOp, Reducer, Join, Normalize,
                        NumericalColumnComparator, StringComparator)
from .transform_ext import (BooleanColumnCondition, FloatColumnCondition, TimeColumnCondition, InfiniteColumnCondition, TrivialColumnCondition,  # noqa: F401,E402
                            StringRegexColumnCondition, SequenceLengthCondition, InvalidNumColumns, FirstDigitTransform, PivotTransform,
                            ReplaceEmptyIntegerWithValue, ParseDoubleTransform, Log2Normalizer, MapAllStringsExceptList, StringListToCategoricalSet,
                            StringListToCountsNDArray, StringListToIndicesNDArray, TextToCharacterIndexTransform, TextToTermIndexSequenceTransform,
                            BaseSequenceExpansionTransform, DeriveColumnsFromTime, NDArrayScalarOpTransform, NDArrayMathFunctionTransform,
                            NDArrayColumnsMathOpTransform, NDArrayDistanceTransform, Distance, SequenceDifferenceTransform, SequenceTrimToLength,
                            ReduceSequenceTransform, SequenceSplitTimeSeparation, TimeWindowFunction, OverlappingTimeWindowFunction,
                            ReduceSequenceByWindowTransform, SequenceMerge, RandomSplit, StringReduceOp, StringReducer)
from .analysis import (AnalyzeLocal, DataAnalysis, SequenceDataAnalysis, SequenceLengthAnalysis, IntegerAnalysis, LongAnalysis,  # noqa: F401,E402,F811
                       DoubleAnalysis, TimeAnalysis, StringAnalysis, CategoricalAnalysis, BytesAnalysis, NDArrayAnalysis,
                       NumericalColumnAnalysis, DataQualityAnalysis, ColumnQuality, DoubleQuality, IntegerQuality, LongQuality,
                       StringQuality, CategoricalQuality, TimeQuality, BytesQuality, HtmlAnalysis, HtmlSequencePlotting,
                       CalculateSortedRank, WritableComparator, DoubleWritableComparator, FloatWritableComparator,
                       IntWritableComparator, LongWritableComparator, TextWritableComparator, ReverseComparator)
