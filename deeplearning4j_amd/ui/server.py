"""UIServer (reference: deeplearning4j-ui-parent/deeplearning4j-vertx/.../ui/VertxUIServer.java + module/train/TrainModule.java:
``UIServer.getInstance().attach(statsStorage)``; routes /train/overview, /train/model, /train/system, /train/sessions,
/remoteReceive).  A stdlib ``ThreadingHTTPServer`` serves the same JSON documents plus a single self-contained HTML page
(inline SVG charts, no external assets)."""
from __future__ import annotations

import json
import threading
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import List, Optional
from urllib.parse import parse_qs, urlparse

from .stats import (InMemoryStatsStorage, Persistable, StatsInitializationReport, StatsReport, StatsStorage, TYPE_ID)

_PAGE = """<!doctype html><html><head><meta charset="utf-8"><title>DL4J-AMD Training UI</title>
<style>body{font-family:sans-serif;margin:20px;background:#fafafa}h1{font-size:20px}.card{background:#fff;border:1px solid #ddd;
border-radius:6px;padding:12px;margin:10px 0}svg{width:100%;height:220px}table{border-collapse:collapse;font-size:13px}
td,th{border-bottom:1px solid #eee;padding:3px 10px;text-align:left}select{margin-left:8px}</style></head><body>
<h1>Training overview <select id="sess"></select></h1>
<div class="card"><b>Model score vs. iteration</b><svg id="score"></svg></div>
<div class="card"><b>log10 update : parameter ratio (mean magnitudes)</b><svg id="ratio"></svg></div>
<div class="card"><b>Performance</b><table id="perf"></table></div>
<div class="card"><b>Model / system</b><table id="info"></table></div>
<script>
function line(svg, series){const W=svg.clientWidth||800,H=220,p=35;svg.innerHTML='';let xs=[],ys=[];
 series.forEach(s=>s.pts.forEach(a=>{xs.push(a[0]);ys.push(a[1]);}));if(!xs.length)return;
 const x0=Math.min(...xs),x1=Math.max(...xs)||1,y0=Math.min(...ys),y1=Math.max(...ys);
 const sx=v=>p+(W-2*p)*(v-x0)/((x1-x0)||1),sy=v=>H-p-(H-2*p)*(v-y0)/((y1-y0)||1);
 const cols=['#1f77b4','#d62728','#2ca02c','#9467bd','#ff7f0e','#8c564b'];
 series.forEach((s,i)=>{const d=s.pts.map((a,j)=>(j?'L':'M')+sx(a[0])+' '+sy(a[1])).join(' ');
  svg.innerHTML+='<path d="'+d+'" fill="none" stroke="'+cols[i%6]+'" stroke-width="1.5"/>';});
 svg.innerHTML+='<text x="4" y="'+(p)+'" font-size="11">'+y1.toPrecision(4)+'</text><text x="4" y="'+(H-p)+'" font-size="11">'+y0.toPrecision(4)+
 '</text><text x="'+p+'" y="'+(H-8)+'" font-size="11">'+x0+'</text><text x="'+(W-p-30)+'" y="'+(H-8)+'" font-size="11">'+x1+'</text>';}
async function refresh(){const s=await (await fetch('/train/sessions')).json();const sel=document.getElementById('sess');
 if(sel.options.length!=s.length){sel.innerHTML=s.map(x=>'<option>'+x+'</option>').join('');}
 const id=sel.value||s[s.length-1];if(!id)return;
 const o=await (await fetch('/train/overview/data?sessionId='+id)).json();
 line(document.getElementById('score'),[{pts:o.scores}]);
 line(document.getElementById('ratio'),Object.keys(o.updateRatios).slice(0,6).map(k=>({pts:o.updateRatios[k]})));
 document.getElementById('perf').innerHTML=Object.entries(o.perf||{}).map(e=>'<tr><th>'+e[0]+'</th><td>'+e[1]+'</td></tr>').join('');
 document.getElementById('info').innerHTML=Object.entries(o.model||{}).map(e=>'<tr><th>'+e[0]+'</th><td>'+e[1]+'</td></tr>').join('');}
refresh();setInterval(refresh,2000);</script></body></html>"""


class UIServer:
    _instance: Optional["UIServer"] = None

    def __init__(self, port=9000, host="127.0.0.1"):
        self.port, self.host = port, host
        self._storages: List[StatsStorage] = []
        self._remote = InMemoryStatsStorage()
        self._remote_enabled = False
        self._httpd: Optional[ThreadingHTTPServer] = None
        self._thread = None

    @staticmethod
    def getInstance(port=9000, host="127.0.0.1") -> "UIServer":
        if UIServer._instance is None or UIServer._instance._httpd is None:
            UIServer._instance = UIServer(port, host)
            UIServer._instance.start()
        return UIServer._instance

    # -- storage management
    def attach(self, storage: StatsStorage):
        if storage not in self._storages:
            self._storages.append(storage)

    def detach(self, storage):
        if storage in self._storages:
            self._storages.remove(storage)

    def isAttached(self, storage):
        return storage in self._storages

    def getStatsStorageInstances(self):
        return list(self._storages)

    def enableRemoteListener(self, storage: Optional[StatsStorage] = None, attach=True):
        self._remote_enabled = True
        if storage is not None:
            self._remote = storage
        if attach:
            self.attach(self._remote)

    def disableRemoteListener(self):
        self._remote_enabled = False

    def isRemoteListenerEnabled(self):
        return self._remote_enabled

    # -- data assembly (TrainModule.getOverviewData / getModelData)
    def _find(self, sessionId):
        for st in self._storages:
            if st.sessionExists(sessionId):
                return st
        return None

    def sessions(self):
        out = []
        for st in self._storages:
            out += st.listSessionIDs()
        return out

    def overview(self, sessionId):
        st = self._find(sessionId)
        if st is None:
            return {"scores": [], "updateRatios": {}, "perf": {}, "model": {}}
        scores, ratios, perf = [], {}, {}
        for w in st.listWorkerIDsForSessionAndType(sessionId, TYPE_ID):
            for u in st.getAllUpdatesAfter(sessionId, TYPE_ID, w, -1):
                it = u.get("iteration", 0)
                scores.append([it, u.get("score")])
                ps, us = u.get("parameters", {}), u.get("updates", {})
                for k, uv in us.items():
                    pm, um = ps.get(k, {}).get("meanMagnitude"), uv.get("meanMagnitude")
                    if pm and um and pm > 0 and um > 0:
                        import math
                        ratios.setdefault(k, []).append([it, math.log10(um / pm)])
                if u.get("performance"):
                    perf = u.get("performance")
        scores.sort()
        model = {}
        infos = st.getAllStaticInfos(sessionId, TYPE_ID)
        if infos:
            c = infos[0].content
            model = {"class": c["model"]["modelClass"], "parameters": c["model"]["numParams"], "layers": c["model"]["numLayers"],
                     "device": c["hardware"]["deviceName"], "backend": c["software"]["backend"], "torch": c["software"]["torch"]}
        return {"scores": scores, "updateRatios": ratios, "perf": perf, "model": model}

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

    def system(self, sessionId):
        st = self._find(sessionId)
        if st is None:
            return {}
        infos = st.getAllStaticInfos(sessionId, TYPE_ID)
        mem = []
        for w in st.listWorkerIDsForSessionAndType(sessionId, TYPE_ID):
            for u in st.getAllUpdatesAfter(sessionId, TYPE_ID, w, -1):
                if u.get("memory"):
                    mem.append([u.get("iteration"), u["memory"]])
        return {"hardware": infos[0]["hardware"] if infos else {}, "software": infos[0]["software"] if infos else {}, "memory": mem}

    def _receive(self, d):
        if not self._remote_enabled:
            return False
        cls = {"StatsInitializationReport": StatsInitializationReport, "StatsReport": StatsReport}.get(d.get("class"), Persistable)
        p = cls(d["sessionID"], d["typeID"], d["workerID"], d["timestamp"], d["content"])
        {"meta": self._remote.putStorageMetaData, "static": self._remote.putStaticInfo, "update": self._remote.putUpdate}[d["type"]](p)
        return True

    # -- http
    def start(self):
        ui = self

        class H(BaseHTTPRequestHandler):
            def log_message(self, *a):
                pass

            def _json(self, obj, code=200):
                body = json.dumps(obj).encode()
                self.send_response(code)
                self.send_header("Content-Type", "application/json")
                self.send_header("Content-Length", str(len(body)))
                self.end_headers()
                self.wfile.write(body)

            def do_GET(self):
                u = urlparse(self.path)
                q = {k: v[0] for k, v in parse_qs(u.query).items()}
                sid = q.get("sessionId") or (ui.sessions()[-1] if ui.sessions() else None)
                if u.path in ("/", "/train", "/train/overview"):
                    body = _PAGE.encode()
                    self.send_response(200)
                    self.send_header("Content-Type", "text/html; charset=utf-8")
                    self.send_header("Content-Length", str(len(body)))
                    self.end_headers()
                    self.wfile.write(body)
                elif u.path == "/train/sessions":
                    self._json(ui.sessions())
                elif u.path == "/train/overview/data":
                    self._json(ui.overview(sid))
                elif u.path == "/train/model/data":
                    self._json(ui.model_data(sid, q.get("param")))
                elif u.path == "/train/system/data":
                    self._json(ui.system(sid))
                else:
                    self._json({"error": "not found"}, 404)

            def do_POST(self):
                if urlparse(self.path).path != "/remoteReceive":
                    return self._json({"error": "not found"}, 404)
                n = int(self.headers.get("Content-Length", "0"))
                ok = ui._receive(json.loads(self.rfile.read(n).decode()))
                self._json({"ok": ok}, 200 if ok else 403)

        self._httpd = ThreadingHTTPServer((self.host, self.port), H)
        self.port = self._httpd.server_address[1]
        self._thread = threading.Thread(target=self._httpd.serve_forever, daemon=True, name="UIServer")
        self._thread.start()
        return self

    def getPort(self):
        return self.port

    def getAddress(self):
        return "http://%s:%d" % (self.host, self.port)

    def stop(self):
        if self._httpd is not None:
            self._httpd.shutdown()
            self._httpd.server_close()
            self._httpd = None
        if UIServer._instance is self:
            UIServer._instance = None
