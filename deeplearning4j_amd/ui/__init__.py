"""Training UI (reference: deeplearning4j-ui-parent/**: deeplearning4j-ui-model {StatsListener, StatsStorage, InMemoryStatsStorage,
FileStatsStorage, StatsInitializationReport, StatsReport}, deeplearning4j-vertx {UIServer, RemoteUIStatsStorageRouter})."""
from .stats import (StatsListener, StatsUpdateConfiguration, StatsInitializationReport, StatsReport, StatsStorage,  # noqa: F401
                    InMemoryStatsStorage, FileStatsStorage, StatsStorageEvent, RemoteUIStatsStorageRouter)
from .server import UIServer  # noqa: F401
