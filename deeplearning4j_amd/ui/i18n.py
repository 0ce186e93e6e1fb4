"""UI internationalisation (reference: deeplearning4j-ui-parent/deeplearning4j-vertx/src/main/java/org/deeplearning4j/ui/i18n/
{DefaultI18N,I18NProvider,I18NResource}.java and the ``dl4j_i18n/*.<lang>`` resource files): ``getMessage(key)`` /
``getMessage(lang, key)``, a process-wide default language switched by the ``/setlang/<code>`` route, fall back to English for a missing
key. Message keys follow the reference's ``train.<page>.<item>`` naming; the tables are in-line instead of resource files."""
from __future__ import annotations

import threading
from typing import Dict, Optional

DEFAULT_LANGUAGE = "en"

_MESSAGES: Dict[str, Dict[str, str]] = {
    "en": {
        "train.pagetitle": "DL4J-AMD Training UI", "train.nav.overview": "Overview", "train.nav.model": "Model", "train.nav.system": "System",
        "train.nav.language": "Language", "train.session.label": "Session", "train.session.worker.label": "Worker",
        "train.overview.chart.scoreTitle": "Model score vs. iteration", "train.overview.chart.updateRatioTitle": "Update : parameter ratios (log10)",
        "train.overview.chart.stdevTitle": "Standard deviations (log10)", "train.overview.perftable.title": "Performance",
        "train.overview.modeltable.title": "Model and training information", "train.overview.perftable.startTime": "Start time",
        "train.overview.perftable.totalRuntime": "Total runtime", "train.overview.perftable.lastUpdate": "Last update",
        "train.overview.perftable.totalParamUpdates": "Total parameter updates", "train.overview.perftable.updatesPerSec": "Updates / sec",
        "train.overview.perftable.examplesPerSec": "Examples / sec", "train.overview.modeltable.modeltype": "Model type",
        "train.overview.modeltable.nLayers": "Layers", "train.overview.modeltable.nParams": "Total parameters",
        "train.model.layerInfoTable.title": "Layer information", "train.model.meanmag.title": "Mean magnitudes (log10)",
        "train.model.activations.title": "Activations", "train.model.paramHist.title": "Parameter histogram",
        "train.model.updateHist.title": "Update histogram", "train.model.lrChart.title": "Learning rate",
        "train.model.noLayerSelected": "Select a layer", "train.system.title": "System", "train.system.hwTable.title": "Hardware",
        "train.system.swTable.title": "Software", "train.system.chart.deviceMemoryTitle": "Device memory utilisation",
        "train.system.chart.hostMemoryTitle": "Host memory", "tsne.title": "t-SNE", "activations.title": "Convolutional activations",
        "samediff.title": "SameDiff graph",
    },
    "de": {
        "train.nav.overview": "Übersicht", "train.nav.model": "Modell", "train.nav.system": "System", "train.nav.language": "Sprache",
        "train.session.label": "Sitzung", "train.overview.chart.scoreTitle": "Modell-Score pro Iteration",
        "train.overview.chart.updateRatioTitle": "Verhältnis Update : Parameter (log10)", "train.overview.perftable.title": "Leistung",
        "train.overview.modeltable.title": "Modell- und Trainingsinformationen", "train.model.layerInfoTable.title": "Schichtinformationen",
        "train.system.hwTable.title": "Hardware", "train.system.swTable.title": "Software",
    },
    "ja": {
        "train.nav.overview": "概要", "train.nav.model": "モデル", "train.nav.system": "システム", "train.nav.language": "言語",
        "train.session.label": "セッション", "train.overview.chart.scoreTitle": "モデルスコアと反復回数",
        "train.overview.chart.updateRatioTitle": "更新とパラメータの比率 (log10)", "train.overview.perftable.title": "パフォーマンス",
        "train.overview.modeltable.title": "モデルと訓練の情報", "train.model.layerInfoTable.title": "レイヤー情報",
    },
    "zh": {
        "train.nav.overview": "概览", "train.nav.model": "模型", "train.nav.system": "系统", "train.nav.language": "语言",
        "train.session.label": "会话", "train.overview.chart.scoreTitle": "模型得分与迭代次数", "train.overview.perftable.title": "性能",
        "train.overview.modeltable.title": "模型与训练信息", "train.model.layerInfoTable.title": "层信息",
    },
    "ko": {
        "train.nav.overview": "개요", "train.nav.model": "모델", "train.nav.system": "시스템", "train.nav.language": "언어",
        "train.overview.chart.scoreTitle": "모델 점수 대 반복", "train.overview.perftable.title": "성능",
    },
    "ru": {
        "train.nav.overview": "Обзор", "train.nav.model": "Модель", "train.nav.system": "Система", "train.nav.language": "Язык",
        "train.overview.chart.scoreTitle": "Оценка модели по итерациям", "train.overview.perftable.title": "Производительность",
    },
}


class DefaultI18N:
    """``I18N`` implementation: one default-language instance plus per-session instances (I18NProvider)."""
    _instance: Optional["DefaultI18N"] = None
    _sessions: Dict[str, "DefaultI18N"] = {}
    _lock = threading.Lock()

    def __init__(self, language=DEFAULT_LANGUAGE):
        self._lang = language if language in _MESSAGES else DEFAULT_LANGUAGE

    @classmethod
    def getInstance(cls, sessionId: Optional[str] = None) -> "DefaultI18N":
        with cls._lock:
            if sessionId is None:
                if cls._instance is None:
                    cls._instance = DefaultI18N()
                return cls._instance
            if sessionId not in cls._sessions:
                cls._sessions[sessionId] = DefaultI18N(cls._instance._lang if cls._instance else DEFAULT_LANGUAGE)
            return cls._sessions[sessionId]

    @classmethod
    def removeInstance(cls, sessionId):
        with cls._lock:
            return cls._sessions.pop(sessionId, None)

    @staticmethod
    def languages():
        return sorted(_MESSAGES)

    def getDefaultLanguage(self):
        return self._lang

    def setDefaultLanguage(self, code):
        if code not in _MESSAGES:
            raise ValueError("unknown UI language %r (have %s)" % (code, ", ".join(self.languages())))
        self._lang = code

    def getMessage(self, *a):
        """getMessage(key) or getMessage(langCode, key); English when the language has no entry, the key itself when nobody has."""
        lang, key = (self._lang, a[0]) if len(a) == 1 else a
        return _MESSAGES.get(lang, {}).get(key) or _MESSAGES[DEFAULT_LANGUAGE].get(key) or key

    def getMessages(self, langCode=None):
        out = dict(_MESSAGES[DEFAULT_LANGUAGE])
        out.update(_MESSAGES.get(langCode or self._lang, {}))
        return out


class I18NProvider:
    @staticmethod
    def getInstance(sessionId=None):
        return DefaultI18N.getInstance(sessionId)

    @staticmethod
    def removeInstance(sessionId):
        return DefaultI18N.removeInstance(sessionId)
