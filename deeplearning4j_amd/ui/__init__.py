"""Training UI (reference: deeplearning4j-ui-parent/**: deeplearning4j-ui-model {StatsListener, StatsStorage, InMemoryStatsStorage,
FileStatsStorage, StatsInitializationReport, StatsReport}, deeplearning4j-vertx {UIServer, RemoteUIStatsStorageRouter})."""
from .stats import (StatsListener, StatsUpdateConfiguration, StatsInitializationReport, StatsReport, StatsStorage,  # noqa: F401
                    InMemoryStatsStorage, FileStatsStorage, StatsStorageEvent, RemoteUIStatsStorageRouter)
from .server import UIServer  # noqa: F401
from .modules import (UIModule, Route, HttpMethod, TrainModule, RemoteReceiverModule, TsneModule, ConvolutionalListenerModule,  # noqa: F401
                      SameDiffModule, DefaultModule, buildGraphInfo)
from .i18n import DefaultI18N, I18NProvider  # noqa: F401
from .weights import ConvolutionalIterationListener  # noqa: F401
from . import components  # noqa: F401
