"""UI modules and their routes (reference: deeplearning4j-ui-parent/deeplearning4j-vertx/src/main/java/org/deeplearning4j/ui/
{api/{UIModule,Route,HttpMethod}.java, module/train/{TrainModule,TrainModuleUtils}.java, module/remote/RemoteReceiverModule.java,
module/tsne/TsneModule.java, module/convolutional/ConvolutionalListenerModule.java, module/SameDiffModule.java,
module/defaultModule/DefaultModule.java}).

A module declares ``Route(path, method, handler)`` entries (``:name`` path segments become handler arguments) and the stats type IDs it
wants storage events for; ``UIServer`` dispatches HTTP requests over the registered modules. Pages are self-contained HTML (inline JS +
SVG); every page has a JSON twin, which is what the tests and remote tools read.
"""
from __future__ import annotations

import json
import math
import re
import time
from typing import Callable, Dict, List, Optional

from .i18n import DefaultI18N
from .stats import (InMemoryStatsStorage, Persistable, StatsInitializationReport, StatsReport, StatsStorage, TYPE_ID)


class HttpMethod:
    GET, POST, PUT, DELETE = "GET", "POST", "PUT", "DELETE"


class Route:
    """One HTTP route. ``handler(ctx, *pathParams)`` returns a dict / list (sent as JSON), a str (HTML), or ``(status, body, contentType)``."""

    def __init__(self, route: str, method: str, handler: Callable):
        self.route, self.method, self.handler = route, method, handler
        parts = []
        for seg in route.strip("/").split("/"):
            parts.append("(?P<%s>[^/]+)" % seg[1:] if seg.startswith(":") else re.escape(seg))
        self._re = re.compile("^/" + "/".join(parts) + "/?$")

    def match(self, method, path):
        if method != self.method:
            return None
        m = self._re.match(path)
        return None if m is None else list(m.groups())


class RequestContext:
    def __init__(self, query: Dict[str, str], body: bytes = b"", headers=None):
        self.query, self.body, self.headers = query, body, headers or {}

    def json(self):
        return json.loads(self.body.decode() or "null")


class UIModule:
    def getCallbackTypeIDs(self) -> List[str]:
        return []

    def getRoutes(self) -> List[Route]:
        return []

    def reportStorageEvents(self, events):
        pass

    def onAttach(self, storage: StatsStorage):
        pass

    def onDetach(self, storage: StatsStorage):
        pass

    def getInternationalizationResources(self):
        return []


# ----------------------------------------------------------------------------- shared page shell
_SHELL = """<!doctype html><html><head><meta charset="utf-8"><title>%(title)s</title>
<style>body{font-family:sans-serif;margin:0;background:#fafafa}nav{background:#263238;color:#fff;padding:10px 20px}nav a{color:#cfd8dc;margin-right:18px;
text-decoration:none}nav a.on{color:#fff;font-weight:bold}main{margin:20px}.card{background:#fff;border:1px solid #ddd;border-radius:6px;padding:12px;
margin:10px 0}svg{width:100%%;height:220px}table{border-collapse:collapse;font-size:13px}td,th{border-bottom:1px solid #eee;padding:3px 10px;text-align:left}
select{margin-left:8px}.node{cursor:pointer}.node rect{fill:#e3f2fd;stroke:#1976d2}.node.sel rect{fill:#ffe082}</style></head><body>
<nav><a href="/train/overview" id="n-overview">%(overview)s</a><a href="/train/model" id="n-model">%(model)s</a><a href="/train/system" id="n-system">%(system)s</a>
<a href="/tsne">t-SNE</a><a href="/activations">Activations</a><a href="/samediff">SameDiff</a>
<span style="float:right">%(session)s <select id="sess"></select> %(language)s <select id="lang">%(langs)s</select></span></nav><main id="main"></main>
<script>
const I18N=%(messages)s;function T(k){return I18N[k]||k;}
function line(svg,series){const W=svg.clientWidth||800,H=220,p=38;svg.innerHTML='';let xs=[],ys=[];series.forEach(s=>s.pts.forEach(a=>{if(a[1]!=null&&isFinite(a[1])){xs.push(a[0]);ys.push(a[1]);}}));
 if(!xs.length)return;const x0=Math.min(...xs),x1=Math.max(...xs),y0=Math.min(...ys),y1=Math.max(...ys);const sx=v=>p+(W-2*p)*(v-x0)/((x1-x0)||1),sy=v=>H-p-(H-2*p)*(v-y0)/((y1-y0)||1);
 const cols=['#1f77b4','#d62728','#2ca02c','#9467bd','#ff7f0e','#8c564b','#e377c2','#7f7f7f'];
 series.forEach((s,i)=>{const d=s.pts.filter(a=>a[1]!=null&&isFinite(a[1])).map((a,j)=>(j?'L':'M')+sx(a[0])+' '+sy(a[1])).join(' ');
  svg.innerHTML+='<path d="'+d+'" fill="none" stroke="'+cols[i%%8]+'" stroke-width="1.5"/>'+(s.name?'<text x="'+(W-p-140)+'" y="'+(14+12*i)+'" font-size="10" fill="'+cols[i%%8]+'">'+s.name+'</text>':'');});
 svg.innerHTML+='<text x="2" y="'+p+'" font-size="10">'+y1.toPrecision(4)+'</text><text x="2" y="'+(H-p)+'" font-size="10">'+y0.toPrecision(4)+'</text><text x="'+p+'" y="'+(H-6)+'" font-size="10">'+x0+
 '</text><text x="'+(W-p-30)+'" y="'+(H-6)+'" font-size="10">'+x1+'</text>';}
function hist(svg,h){svg.innerHTML='';if(!h||!h.counts||!h.counts.length)return;const W=svg.clientWidth||800,H=220,p=38,n=h.counts.length,m=Math.max(...h.counts,1);
 h.counts.forEach((c,i)=>{const x=p+(W-2*p)*i/n,hh=(H-2*p)*c/m;svg.innerHTML+='<rect x="'+x+'" y="'+(H-p-hh)+'" width="'+Math.max((W-2*p)/n-1,1)+'" height="'+hh+'" fill="#1f77b4"/>';});
 svg.innerHTML+='<text x="'+p+'" y="'+(H-6)+'" font-size="10">'+(+h.min).toPrecision(4)+'</text><text x="'+(W-p-40)+'" y="'+(H-6)+'" font-size="10">'+(+h.max).toPrecision(4)+'</text>';}
function table(el,obj){el.innerHTML=Object.entries(obj||{}).map(e=>'<tr><th>'+e[0]+'</th><td>'+(e[1]==null?'':e[1])+'</td></tr>').join('');}
async function J(u){return await (await fetch(u)).json();}
let SID=null;
async function sessions(){const s=await J('/train/sessions/all');const sel=document.getElementById('sess');if(sel.options.length!=s.length)sel.innerHTML=s.map(x=>'<option>'+x+'</option>').join('');
 SID=sel.value||s[s.length-1]||null;return SID;}
document.getElementById('lang').onchange=async e=>{await fetch('/setlang/'+e.target.value);location.reload();};
%(script)s
</script></body></html>"""

_OVERVIEW_JS = """
document.getElementById('n-overview').className='on';
document.getElementById('main').innerHTML='<div class="card"><b>'+T('train.overview.chart.scoreTitle')+'</b><svg id="score"></svg></div><div class="card"><b>'+
 T('train.overview.chart.updateRatioTitle')+'</b><svg id="ratio"></svg></div><div class="card"><b>'+T('train.overview.chart.stdevTitle')+'</b><svg id="stdev"></svg></div><div class="card"><b>'+
 T('train.overview.perftable.title')+'</b><table id="perf"></table></div><div class="card"><b>'+T('train.overview.modeltable.title')+'</b><table id="info"></table></div>';
async function refresh(){const id=await sessions();if(!id)return;const o=await J('/train/overview/data?sessionId='+id);
 line(document.getElementById('score'),[{pts:o.scores}]);
 line(document.getElementById('ratio'),Object.keys(o.updateRatios).slice(0,8).map(k=>({name:k,pts:o.updateRatios[k]})));
 line(document.getElementById('stdev'),Object.keys(o.stdevUpdates||{}).slice(0,8).map(k=>({name:k,pts:o.stdevUpdates[k]})));
 table(document.getElementById('perf'),o.perf);table(document.getElementById('info'),o.model);}
refresh();setInterval(refresh,2000);"""

_MODEL_JS = """
document.getElementById('n-model').className='on';
document.getElementById('main').innerHTML='<div class="card"><svg id="graph" style="height:auto"></svg></div><div class="card"><b>'+T('train.model.layerInfoTable.title')+'</b><table id="linfo"></table></div>'+
 '<div class="card"><b>'+T('train.model.meanmag.title')+'</b><svg id="mm"></svg></div><div class="card"><b>'+T('train.model.lrChart.title')+'</b><svg id="lr"></svg></div>'+
 '<div class="card"><b>'+T('train.model.paramHist.title')+'</b><svg id="ph"></svg></div><div class="card"><b>'+T('train.model.updateHist.title')+'</b><svg id="uh"></svg></div>';
let LAYER=null;
async function graph(){const id=await sessions();if(!id)return;const g=await J('/train/model/graph?sessionId='+id);const svg=document.getElementById('graph');const n=g.vertexNames.length;
 svg.setAttribute('viewBox','0 0 800 '+(n*34+10));svg.style.height=(n*34+10)+'px';let h='';
 g.vertexNames.forEach((v,i)=>{(g.vertexInputs[i]||[]).forEach(j=>{h+='<path d="M'+(60+((j*37)%5)*4)+' '+(j*34+28)+'C20 '+(j*34+40)+',20 '+(i*34)+','+(60+((j*37)%5)*4)+' '+(i*34+6)+'" stroke="#90a4ae" fill="none"/>';});
  h+='<g class="node'+(LAYER==i?' sel':'')+'" onclick="LAYER='+i+';graph();layer();"><rect x="60" y="'+(i*34+6)+'" width="360" height="22" rx="4"/><text x="68" y="'+(i*34+21)+'" font-size="12">'+v+' ('+g.vertexTypes[i]+')</text></g>';});
 svg.innerHTML=h;}
async function layer(){if(LAYER==null||!SID)return;const d=await J('/train/model/data/'+LAYER+'?sessionId='+SID);table(document.getElementById('linfo'),d.layerInfo);
 line(document.getElementById('mm'),Object.keys(d.meanMag||{}).map(k=>({name:k,pts:d.meanMag[k]})));line(document.getElementById('lr'),[{pts:d.learningRates||[]}]);
 hist(document.getElementById('ph'),Object.values(d.paramHist||{})[0]);hist(document.getElementById('uh'),Object.values(d.updateHist||{})[0]);}
graph();setInterval(()=>{graph();layer();},3000);"""

_SYSTEM_JS = """
document.getElementById('n-system').className='on';
document.getElementById('main').innerHTML='<div class="card"><b>'+T('train.system.chart.deviceMemoryTitle')+'</b><svg id="dm"></svg></div><div class="card"><b>'+T('train.system.chart.hostMemoryTitle')+
 '</b><svg id="hm"></svg></div><div class="card"><b>'+T('train.system.hwTable.title')+'</b><table id="hw"></table></div><div class="card"><b>'+T('train.system.swTable.title')+'</b><table id="sw"></table></div>';
async function refresh(){const id=await sessions();if(!id)return;const o=await J('/train/system/data?sessionId='+id);table(document.getElementById('hw'),o.hardware);table(document.getElementById('sw'),o.software);
 const m=o.memory||[];line(document.getElementById('dm'),[{name:'allocated',pts:m.map(a=>[a[0],a[1].deviceCurrentBytes])},{name:'reserved',pts:m.map(a=>[a[0],a[1].deviceReservedBytes])}]);
 line(document.getElementById('hm'),[{name:'max RSS (kB)',pts:m.map(a=>[a[0],a[1].hostMaxRssKb])}]);}
refresh();setInterval(refresh,2000);"""


def render_page(script, title_key="train.pagetitle"):
    i18n = DefaultI18N.getInstance()
    langs = "".join("<option%s>%s</option>" % (" selected" if l == i18n.getDefaultLanguage() else "", l) for l in DefaultI18N.languages())
    return _SHELL % {"title": i18n.getMessage(title_key), "overview": i18n.getMessage("train.nav.overview"), "model": i18n.getMessage("train.nav.model"),
                     "system": i18n.getMessage("train.nav.system"), "session": i18n.getMessage("train.session.label"),
                     "language": i18n.getMessage("train.nav.language"), "langs": langs, "messages": json.dumps(i18n.getMessages()), "script": script}


# ----------------------------------------------------------------------------- TrainModuleUtils
class GraphInfo:
    def __init__(self, vertexNames, vertexTypes, vertexInputs, vertexInfo, originalVertexName):
        self.vertexNames, self.vertexTypes, self.vertexInputs, self.vertexInfo, self.originalVertexName = (
            vertexNames, vertexTypes, vertexInputs, vertexInfo, originalVertexName)

    def toDict(self):
        return {"vertexNames": self.vertexNames, "vertexTypes": self.vertexTypes, "vertexInputs": self.vertexInputs, "vertexInfo": self.vertexInfo,
                "originalVertexName": self.originalVertexName}


def _layer_info(layer_conf) -> Dict[str, str]:
    out = {}
    for k in ("nIn", "nOut", "kernelSize", "stride", "padding", "activation", "poolingType", "lossFunction", "dropOut"):
        v = getattr(layer_conf, k, None)
        if v is not None and not callable(v):
            out[k] = str(v)
    return out


def buildGraphInfo(model) -> GraphInfo:
    """Vertex list of a network for the model page (TrainModuleUtils.buildGraphInfo): names, types, input indices and a few layer
    facts; index 0 is the input for a MultiLayerNetwork, the network inputs come first for a ComputationGraph."""
    conf = model.conf
    names, types, inputs, info, orig = [], [], [], [], []
    if hasattr(conf, "vertexInputs"):                      # ComputationGraph
        for n in conf.networkInputs:
            names.append(n)
            types.append("Input")
            inputs.append([])
            info.append({})
            orig.append(n)
        index = {n: i for i, n in enumerate(names)}
        order = list(getattr(model, "topologicalOrderNames", None) or conf.vertices.keys())
        for n in order:
            if n in index or n not in conf.vertices:
                continue
            index[n] = len(names)
            v = conf.vertices[n]
            lc = getattr(getattr(v, "layerConf", None), "layer", None)
            names.append(n)
            types.append(type(lc).__name__ if lc is not None else type(v).__name__)
            inputs.append(list(conf.vertexInputs.get(n, [])))
            info.append(_layer_info(lc) if lc is not None else {})
            orig.append(n)
        inputs = [[index[x] for x in ins if x in index] for ins in inputs]
    else:
        names.append("input")
        types.append("Input")
        inputs.append([])
        info.append({})
        orig.append(None)
        for i, l in enumerate(model._ordered_layers()):
            lc = l.conf
            names.append(getattr(lc, "layerName", None) or "layer%d" % i)
            types.append(type(lc).__name__)
            inputs.append([i])
            info.append(_layer_info(lc))
            orig.append(str(i))
    return GraphInfo(names, types, inputs, info, orig)


# ----------------------------------------------------------------------------- TrainModule
def _log10_ratio(a, b):
    return math.log10(a / b) if (a and b and a > 0 and b > 0) else None


class TrainModule(UIModule):
    """Overview / model / system pages over the StatsListener records of the attached storages."""

    def __init__(self):
        self._storages: List[StatsStorage] = []
        self.currentSessionID: Optional[str] = None

    def getCallbackTypeIDs(self):
        return [TYPE_ID]

    def onAttach(self, storage):
        if storage not in self._storages:
            self._storages.append(storage)

    def onDetach(self, storage):
        if storage in self._storages:
            self._storages.remove(storage)

    def reportStorageEvents(self, events):
        for e in events:
            if getattr(e, "typeID", TYPE_ID) == TYPE_ID:
                self.currentSessionID = e.sessionID

    def getRoutes(self):
        G = HttpMethod.GET
        q = lambda ctx: ctx.query.get("sessionId") or self._default_session()
        return [
            Route("/train", G, lambda ctx: (302, "/train/overview", "redirect")),
            Route("/train/overview", G, lambda ctx: render_page(_OVERVIEW_JS)),
            Route("/train/overview/data", G, lambda ctx: self.overview(q(ctx))),
            Route("/train/model", G, lambda ctx: render_page(_MODEL_JS)),
            Route("/train/model/graph", G, lambda ctx: self.model_graph(q(ctx))),
            Route("/train/model/data", G, lambda ctx: self.model_data(q(ctx), ctx.query.get("param"))),
            Route("/train/model/data/:layerId", G, lambda ctx, layerId: self.layer_data(q(ctx), layerId)),
            Route("/train/system", G, lambda ctx: render_page(_SYSTEM_JS)),
            Route("/train/system/data", G, lambda ctx: self.system(q(ctx))),
            Route("/train/sessions", G, lambda ctx: self.sessions()),
            Route("/train/sessions/all", G, lambda ctx: self.sessions()),
            Route("/train/sessions/current", G, lambda ctx: {"sessionId": self._default_session()}),
            Route("/train/sessions/info", G, lambda ctx: self.session_info()),
            Route("/train/:sessionId/info", G, lambda ctx, sessionId: self.session_info().get(sessionId, {})),
            Route("/train/sessions/lastUpdate/:sessionId", G, lambda ctx, sessionId: {"lastUpdate": self.last_update(sessionId)}),
            Route("/train/workers/:sessionId", G, lambda ctx, sessionId: self.workers(sessionId)),
            Route("/setlang/:to", G, lambda ctx, to: self.set_language(to)),
        ]

    # -- helpers
    def _find(self, sessionId):
        for st in self._storages:
            if sessionId is not None and st.sessionExists(sessionId):
                return st
        return None

    def _default_session(self):
        s = self.sessions()
        if self.currentSessionID in s:
            return self.currentSessionID
        return s[-1] if s else None

    def sessions(self):
        out = []
        for st in self._storages:
            out += [s for s in st.listSessionIDs() if TYPE_ID in st.listTypeIDsForSession(s) and s not in out]
        return out

    def workers(self, sessionId):
        st = self._find(sessionId)
        return [] if st is None else st.listWorkerIDsForSessionAndType(sessionId, TYPE_ID)

    def last_update(self, sessionId):
        st = self._find(sessionId)
        if st is None:
            return -1
        ts = [u.timestamp for u in st.getLatestUpdateAllWorkers(sessionId, TYPE_ID) if u is not None]
        return max(ts) if ts else -1

    def session_info(self):
        out = {}
        for sid in self.sessions():
            st = self._find(sid)
            infos = st.getAllStaticInfos(sid, TYPE_ID)
            m = infos[0]["model"] if infos else {}
            out[sid] = {"numWorkers": len(self.workers(sid)), "workers": self.workers(sid), "initTime": infos[0].timestamp if infos else None,
                        "lastUpdate": self.last_update(sid), "modelType": m.get("modelClass"), "numLayers": m.get("numLayers"), "numParams": m.get("numParams")}
        return out

    def set_language(self, to):
        DefaultI18N.getInstance().setDefaultLanguage(to)
        return {"language": to}

    def _updates(self, st, sessionId):
        for w in st.listWorkerIDsForSessionAndType(sessionId, TYPE_ID):
            for u in st.getAllUpdatesAfter(sessionId, TYPE_ID, w, -1):
                yield u

    # -- pages' data
    def overview(self, sessionId):
        st = self._find(sessionId)
        if st is None:
            return {"scores": [], "updateRatios": {}, "stdevParameters": {}, "stdevGradients": {}, "stdevUpdates": {}, "perf": {}, "model": {}}
        scores, ratios, perf = [], {}, {}
        stdev = {"parameters": {}, "gradients": {}, "updates": {}}
        first_ts, last_ts = None, None
        for u in self._updates(st, sessionId):
            it = u.get("iteration", 0)
            scores.append([it, u.get("score")])
            first_ts = u.timestamp if first_ts is None else min(first_ts, u.timestamp)
            last_ts = u.timestamp if last_ts is None else max(last_ts, u.timestamp)
            ps, us = u.get("parameters", {}), u.get("updates", {})
            for k, uv in us.items():
                r = _log10_ratio(uv.get("meanMagnitude"), ps.get(k, {}).get("meanMagnitude"))
                if r is not None:
                    ratios.setdefault(k, []).append([it, r])
            for grp in stdev:
                for k, sv in (u.get(grp) or {}).items():
                    s = sv.get("stdev")
                    if s and s > 0:
                        stdev[grp].setdefault(k, []).append([it, math.log10(s)])
            if u.get("performance"):
                perf = dict(u.get("performance"))
        scores.sort()
        if first_ts is not None:
            perf["startTime"] = time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(first_ts / 1000))
            perf["lastUpdate"] = time.strftime("%Y-%m-%d %H:%M:%S", time.localtime(last_ts / 1000))
        model = {}
        infos = st.getAllStaticInfos(sessionId, TYPE_ID)
        if infos:
            c = infos[0].content
            model = {"class": c["model"]["modelClass"], "parameters": c["model"]["numParams"], "layers": c["model"]["numLayers"],
                     "device": c["hardware"]["deviceName"], "backend": c["software"]["backend"], "torch": c["software"]["torch"]}
        return {"scores": scores, "updateRatios": ratios, "stdevParameters": stdev["parameters"], "stdevGradients": stdev["gradients"],
                "stdevUpdates": stdev["updates"], "perf": perf, "model": model}

    def model_graph(self, sessionId):
        st = self._find(sessionId)
        infos = st.getAllStaticInfos(sessionId, TYPE_ID) if st is not None else []
        g = infos[0]["model"].get("graph") if infos else None
        if g:
            return g
        names = ["input"] + list(infos[0]["model"].get("layerNames", [])) if infos else []
        return {"vertexNames": names, "vertexTypes": ["Input"] + ["Layer"] * (len(names) - 1) if names else [],
                "vertexInputs": [[]] + [[i] for i in range(len(names) - 1)] if names else [], "vertexInfo": [{} for _ in names],
                "originalVertexName": [None] + [str(i) for i in range(len(names) - 1)] if names else []}

    def model_data(self, sessionId, param=None):
        st = self._find(sessionId)
        out = {"paramNames": [], "latest": {}}
        if st is None:
            return out
        for w in st.listWorkerIDsForSessionAndType(sessionId, TYPE_ID):
            u = st.getLatestUpdate(sessionId, TYPE_ID, w)
            if u is None:
                continue
            ps = u.get("parameters", {})
            out["paramNames"] = list(ps)
            for k in ([param] if param else list(ps)):
                out["latest"][k] = {"parameters": ps.get(k), "gradients": u.get("gradients", {}).get(k), "updates": u.get("updates", {}).get(k)}
        return out

    def layer_data(self, sessionId, layerId):
        """Everything the model page shows for one vertex: layer facts, mean-magnitude history of its parameters and updates, learning-rate
        history, latest parameter / update histograms (TrainModule.getModelData)."""
        st = self._find(sessionId)
        out = {"layerInfo": {}, "meanMag": {}, "learningRates": [], "paramHist": {}, "updateHist": {}, "paramNames": []}
        if st is None:
            return out
        g = self.model_graph(sessionId)
        try:
            idx = int(layerId)
        except ValueError:
            idx = g["vertexNames"].index(layerId) if layerId in g["vertexNames"] else -1
        if not 0 <= idx < len(g["vertexNames"]):
            return out
        key = g["originalVertexName"][idx]
        out["layerInfo"] = dict({"name": g["vertexNames"][idx], "type": g["vertexTypes"][idx]}, **(g["vertexInfo"][idx] or {}))
        if key is None:
            return out
        prefix = key + "_"
        latest = None
        for u in self._updates(st, sessionId):
            it = u.get("iteration", 0)
            for grp, label in (("parameters", "param"), ("updates", "update")):
                for k, sv in (u.get(grp) or {}).items():
                    if k.startswith(prefix) and sv.get("meanMagnitude"):
                        out["meanMag"].setdefault("%s %s" % (label, k[len(prefix):]), []).append([it, math.log10(max(sv["meanMagnitude"], 1e-30))])
            lr = (u.get("learningRates") or {}).get(key)
            if lr is not None:
                out["learningRates"].append([it, lr])
            latest = u if latest is None or u.timestamp >= latest.timestamp else latest
        if latest is not None:
            for grp, dst in (("parameters", "paramHist"), ("updates", "updateHist")):
                for k, sv in (latest.get(grp) or {}).items():
                    if k.startswith(prefix):
                        if k[len(prefix):] not in out["paramNames"]:
                            out["paramNames"].append(k[len(prefix):])
                        if sv.get("hist") is not None:
                            out[dst][k[len(prefix):]] = {"min": sv.get("min"), "max": sv.get("max"), "counts": sv.get("hist")}
        return out

    def system(self, sessionId):
        st = self._find(sessionId)
        if st is None:
            return {}
        infos = st.getAllStaticInfos(sessionId, TYPE_ID)
        mem = [[u.get("iteration"), u["memory"]] for u in self._updates(st, sessionId) if u.get("memory")]
        return {"hardware": infos[0]["hardware"] if infos else {}, "software": infos[0]["software"] if infos else {}, "memory": mem}


# ----------------------------------------------------------------------------- RemoteReceiverModule
class RemoteReceiverModule(UIModule):
    """POST /remoteReceive: stores reports sent by RemoteUIStatsStorageRouter (RemoteReceiverModule.java)."""

    def __init__(self):
        self.enabled = False
        self.storage: StatsStorage = InMemoryStatsStorage()

    def setEnabled(self, v):
        self.enabled = bool(v)

    def isEnabled(self):
        return self.enabled

    def setStatsStorage(self, st):
        self.storage = st

    def getRoutes(self):
        return [Route("/remoteReceive", HttpMethod.POST, self._post)]

    def receive(self, d):
        if not self.enabled:
            return False
        cls = {"StatsInitializationReport": StatsInitializationReport, "StatsReport": StatsReport}.get(d.get("class"), Persistable)
        p = cls(d["sessionID"], d["typeID"], d["workerID"], d["timestamp"], d["content"])
        {"meta": self.storage.putStorageMetaData, "static": self.storage.putStaticInfo, "update": self.storage.putUpdate}[d["type"]](p)
        return True

    def _post(self, ctx):
        ok = self.receive(ctx.json())
        return (200 if ok else 403, json.dumps({"ok": ok}), "application/json")


# ----------------------------------------------------------------------------- TsneModule
_TSNE_JS = """
document.getElementById('main').innerHTML='<div class="card"><b>t-SNE</b> <select id="ts"></select> <input type="file" id="up"><svg id="plot" style="height:600px"></svg></div>';
async function draw(){const s=await J('/tsne/sessions');const sel=document.getElementById('ts');if(sel.options.length!=s.length)sel.innerHTML=s.map(x=>'<option>'+x+'</option>').join('');
 const id=sel.value||s[0];if(!id)return;const rows=await J('/tsne/coords/'+id);const pts=rows.map(r=>r.split(',')).filter(r=>r.length>=2);if(!pts.length)return;
 const xs=pts.map(p=>+p[0]),ys=pts.map(p=>+p[1]);const x0=Math.min(...xs),x1=Math.max(...xs),y0=Math.min(...ys),y1=Math.max(...ys);const svg=document.getElementById('plot');const W=svg.clientWidth||800,H=600;
 svg.innerHTML=pts.map(p=>{const x=20+(W-40)*(p[0]-x0)/((x1-x0)||1),y=H-20-(H-40)*(p[1]-y0)/((y1-y0)||1);return '<circle cx="'+x+'" cy="'+y+'" r="2" fill="#1f77b4"/>'+(p[2]?'<text x="'+(x+3)+'" y="'+y+'" font-size="9">'+p[2]+'</text>':'');}).join('');}
document.getElementById('up').onchange=async e=>{const t=await e.target.files[0].text();await fetch('/tsne/upload',{method:'POST',body:t});draw();};
document.getElementById('ts').onchange=draw;draw();"""


class TsneModule(UIModule):
    """t-SNE coordinate viewer: rows are ``x,y[,label]`` strings (TsneModule.java: /tsne, /tsne/sessions, /tsne/coords/:sid, /tsne/upload,
    /tsne/post/:sid)."""
    UPLOADED_FILE = "UploadedFile"

    def __init__(self):
        self.knownSessionIDs: Dict[str, List[str]] = {}

    def getRoutes(self):
        G, P = HttpMethod.GET, HttpMethod.POST
        return [Route("/tsne", G, lambda ctx: render_page(_TSNE_JS, "tsne.title")),
                Route("/tsne/sessions", G, lambda ctx: sorted(self.knownSessionIDs)),
                Route("/tsne/coords/:sid", G, lambda ctx, sid: self.knownSessionIDs.get(sid, [])),
                Route("/tsne/upload", P, lambda ctx: self._store(self.UPLOADED_FILE, ctx.body.decode())),
                Route("/tsne/post/:sid", P, lambda ctx, sid: self._store(sid, ctx.body.decode()))]

    def _store(self, sid, text):
        rows = [ln.strip() for ln in text.splitlines() if ln.strip()]
        self.knownSessionIDs[sid] = rows
        return {"status": "ok", "sessionId": sid, "rows": len(rows)}

    def setCoordinates(self, sid, coords, labels=None):
        """Programmatic upload: ``coords`` [n, 2+] array-like, optional labels."""
        rows = []
        for i, c in enumerate(coords):
            row = ",".join("%g" % float(v) for v in list(c)[:3 if len(c) > 2 and labels is None else 2])
            rows.append(row + ("," + str(labels[i]) if labels is not None else ""))
        self.knownSessionIDs[sid] = rows


# ----------------------------------------------------------------------------- ConvolutionalListenerModule
_ACT_JS = """
document.getElementById('main').innerHTML='<div class="card"><b>'+T('activations.title')+'</b><div id="meta"></div><img id="img" style="max-width:100%;image-rendering:pixelated"></div>';
async function refresh(){const m=await J('/activations/info');document.getElementById('meta').textContent=m.iteration==null?'no data':'iteration '+m.iteration+' ('+m.layers.join(', ')+')';
 if(m.iteration!=null)document.getElementById('img').src='/activations/data?t='+Date.now();}
refresh();setInterval(refresh,3000);"""


class ConvolutionalListenerModule(UIModule):
    """Shows the latest activation-grid image posted by ConvolutionalIterationListener (ConvolutionalListenerModule.java: /activations,
    /activations/data)."""
    TYPE_ID = "ConvolutionalListener"

    def __init__(self):
        self._storages: List[StatsStorage] = []
        self.lastSessionID = None
        self.lastWorkerID = None

    def getCallbackTypeIDs(self):
        return [self.TYPE_ID]

    def onAttach(self, storage):
        if storage not in self._storages:
            self._storages.append(storage)

    def onDetach(self, storage):
        if storage in self._storages:
            self._storages.remove(storage)

    def reportStorageEvents(self, events):
        for e in events:
            if getattr(e, "typeID", None) == self.TYPE_ID:
                self.lastSessionID, self.lastWorkerID = e.sessionID, e.workerID

    def _latest(self):
        best = None
        for st in self._storages:
            for sid in st.listSessionIDs():
                if self.TYPE_ID not in st.listTypeIDsForSession(sid):
                    continue
                for u in st.getLatestUpdateAllWorkers(sid, self.TYPE_ID):
                    if u is not None and (best is None or u.timestamp >= best.timestamp):
                        best = u
        return best

    def getRoutes(self):
        G = HttpMethod.GET
        return [Route("/activations", G, lambda ctx: render_page(_ACT_JS, "activations.title")),
                Route("/activations/info", G, lambda ctx: self._info()),
                Route("/activations/data", G, lambda ctx: self._image())]

    def _info(self):
        u = self._latest()
        return {"iteration": None, "layers": []} if u is None else {"iteration": u.get("iteration"), "layers": u.get("layers", []),
                                                                    "width": u.get("width"), "height": u.get("height")}

    def _image(self):
        import base64
        u = self._latest()
        if u is None:
            return (404, json.dumps({"error": "no activations received"}), "application/json")
        return (200, base64.b64decode(u.get("png")), "image/png")


# ----------------------------------------------------------------------------- SameDiffModule
_SD_JS = """
document.getElementById('main').innerHTML='<div class="card"><b>'+T('samediff.title')+'</b> <input id="f" size="60" placeholder="path of a SameDiff .fb / log file on the server"><button id="go">load</button><table id="ops"></table></div>';
document.getElementById('go').onclick=async()=>{const g=await J('/samediff/graph?file='+encodeURIComponent(document.getElementById('f').value));
 document.getElementById('ops').innerHTML='<tr><th>#</th><th>op</th><th>name</th><th>inputs</th><th>outputs</th></tr>'+(g.ops||[]).map((o,i)=>'<tr><td>'+i+'</td><td>'+o.op+'</td><td>'+o.name+'</td><td>'+o.inputs.join(', ')+'</td><td>'+o.outputs.join(', ')+'</td></tr>').join('')||('<tr><td>'+(g.error||'')+'</td></tr>');};"""


class SameDiffModule(UIModule):
    """Graph view of a saved SameDiff model (SameDiffModule.java: /samediff, /samediff/setfile, /samediff/fetch): the server loads the
    FlatBuffers file with the in-tree reader and returns variables + ops as JSON."""

    def __init__(self):
        self.currentFile: Optional[str] = None

    def getRoutes(self):
        G = HttpMethod.GET
        return [Route("/samediff", G, lambda ctx: render_page(_SD_JS, "samediff.title")),
                Route("/samediff/setfile", G, lambda ctx: self.setFile(ctx.query.get("file"))),
                Route("/samediff/graph", G, lambda ctx: self.graph(ctx.query.get("file") or self.currentFile))]

    def setFile(self, path):
        self.currentFile = path
        return {"file": path}

    @staticmethod
    def describe(sd) -> dict:
        vs = [{"name": v.name, "type": str(v.variableType), "dtype": str(v.dtype) if v.dtype is not None else None,
               "shape": list(v.shape) if v.shape is not None else None} for v in sd.variables.values()]
        ops = [{"name": o.name, "op": o.opName, "inputs": list(o.inputs), "outputs": list(o.outputs)} for o in sd.ops.values()]
        return {"variables": vs, "ops": ops}

    def graph(self, path):
        if not path:
            return {"error": "no file set", "variables": [], "ops": []}
        try:
            from ..autodiff.samediff import SameDiff
            return self.describe(SameDiff.load(path))
        except Exception as e:  # shown in the page instead of a 500
            return {"error": "%s: %s" % (type(e).__name__, e), "variables": [], "ops": []}


class DefaultModule(UIModule):
    """``/`` -> training overview (DefaultModule.java); ``multiSession`` mode serves the session chooser instead."""

    def __init__(self, multiSession=False):
        self.multiSession = multiSession

    def getRoutes(self):
        return [Route("/", HttpMethod.GET, lambda ctx: (302, "/train/overview", "redirect"))]
