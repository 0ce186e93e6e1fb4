"""Model import (reference: deeplearning4j-modelimport)."""
from .keras import (KerasModelImport, KerasModel, KerasSequentialModel, KerasLayerMapper, InvalidKerasConfigurationException,  # noqa: F401
                    UnsupportedKerasConfigurationException)
from .hdf5 import Hdf5Archive, Hdf5Writer, write_keras_h5  # noqa: F401
