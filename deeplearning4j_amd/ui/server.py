"""UIServer (reference: deeplearning4j-ui-parent/deeplearning4j-vertx/.../ui/VertxUIServer.java + module/train/TrainModule.java:
``UIServer.getInstance().attach(statsStorage)``; routes /train/overview, /train/model, /train/system, /train/sessions,
/remoteReceive, /tsne, /activations, /samediff, /setlang).  A stdlib ``ThreadingHTTPServer`` dispatches over the ``UIModule``s of
``ui/modules.py``, which serve the JSON documents plus self-contained HTML pages (inline SVG charts, no external assets)."""
from __future__ import annotations

import json
import threading
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import List, Optional
from urllib.parse import parse_qs, urlparse

from .modules import (ConvolutionalListenerModule, DefaultModule, RemoteReceiverModule, RequestContext, SameDiffModule, TrainModule, TsneModule,
                      UIModule)
from .stats import InMemoryStatsStorage, StatsStorage

class UIServer:
    """HTTP front end over a list of ``UIModule``s (VertxUIServer.java): requests are matched against the modules' routes, storage events
    are forwarded to the modules that subscribed to the event's type ID. ``getInstance(port, multiSession=...)`` is the process singleton;
    ``multiSession`` only changes the landing page, every session of every attached storage is always selectable."""
    _instance: Optional["UIServer"] = None

    def __init__(self, port=9000, host="127.0.0.1", multiSession=False):
        self.port, self.host, self.multiSession = port, host, multiSession
        self._storages: List[StatsStorage] = []
        self._listeners = {}
        self._httpd: Optional[ThreadingHTTPServer] = None
        self._thread = None
        self.train = TrainModule()
        self.remote = RemoteReceiverModule()
        self.tsne = TsneModule()
        self._modules: List[UIModule] = [DefaultModule(multiSession), self.train, self.remote, self.tsne, ConvolutionalListenerModule(), SameDiffModule()]

    @staticmethod
    def getInstance(port=9000, host="127.0.0.1", multiSession=False) -> "UIServer":
        if UIServer._instance is None or UIServer._instance._httpd is None:
            UIServer._instance = UIServer(port, host, multiSession)
            UIServer._instance.start()
        return UIServer._instance

    def isMultiSession(self):
        return self.multiSession

    # -- modules
    def getModules(self):
        return list(self._modules)

    def registerModule(self, module: UIModule):
        """Add a custom module; its routes take part in dispatch immediately and it sees the storages already attached."""
        self._modules.append(module)
        for st in self._storages:
            module.onAttach(st)

    # -- storage management
    def attach(self, storage: StatsStorage):
        if storage in self._storages:
            return
        self._storages.append(storage)

        def forward(ev, self=self):
            for m in self._modules:
                if ev.typeID in m.getCallbackTypeIDs():
                    m.reportStorageEvents([ev])
        self._listeners[id(storage)] = forward
        storage.registerStatsStorageListener(forward)
        for m in self._modules:
            m.onAttach(storage)

    def detach(self, storage):
        if storage in self._storages:
            self._storages.remove(storage)
            storage.deregisterStatsStorageListener(self._listeners.pop(id(storage), None))
            for m in self._modules:
                m.onDetach(storage)

    def isAttached(self, storage):
        return storage in self._storages

    def getStatsStorageInstances(self):
        return list(self._storages)

    def enableRemoteListener(self, storage: Optional[StatsStorage] = None, attach=True):
        self.remote.setEnabled(True)
        if storage is not None:
            self.remote.setStatsStorage(storage)
        if attach:
            self.attach(self.remote.storage)

    def disableRemoteListener(self):
        self.remote.setEnabled(False)

    def isRemoteListenerEnabled(self):
        return self.remote.isEnabled()

    # -- data assembly kept on the server object for callers of the round-1 API (delegates to TrainModule)
    def sessions(self):
        return self.train.sessions()

    def overview(self, sessionId):
        return self.train.overview(sessionId)

    def model_data(self, sessionId, param=None):
        return self.train.model_data(sessionId, param)

    def system(self, sessionId):
        return self.train.system(sessionId)

    def _receive(self, d):
        return self.remote.receive(d)

    def dispatch(self, method, path, query, body=b"", headers=None):
        """(status, body bytes, content type[, location]) for one request; also usable without a socket (tests, embedding)."""
        ctx = RequestContext(query, body, headers)
        for m in self._modules:
            for r in m.getRoutes():
                args = r.match(method, path)
                if args is None:
                    continue
                out = r.handler(ctx, *args)
                if isinstance(out, tuple):
                    status, payload, ctype = out
                    if ctype == "redirect":
                        return status, b"", "text/plain", payload
                    return status, payload if isinstance(payload, bytes) else str(payload).encode(), ctype
                if isinstance(out, str):
                    return 200, out.encode(), "text/html; charset=utf-8"
                return 200, json.dumps(out).encode(), "application/json"
        return 404, json.dumps({"error": "not found", "path": path}).encode(), "application/json"

    # -- http
    def start(self):
        ui = self

        class H(BaseHTTPRequestHandler):
            def log_message(self, *a):
                pass

            def _serve(self, method):
                u = urlparse(self.path)
                q = {k: v[0] for k, v in parse_qs(u.query).items()}
                n = int(self.headers.get("Content-Length", "0") or 0)
                body = self.rfile.read(n) if n else b""
                try:
                    res = ui.dispatch(method, u.path, q, body, dict(self.headers))
                except Exception as e:  # a handler bug must not kill the connection thread silently
                    res = (500, json.dumps({"error": "%s: %s" % (type(e).__name__, e)}).encode(), "application/json")
                self.send_response(res[0])
                if len(res) > 3:
                    self.send_header("Location", res[3])
                self.send_header("Content-Type", res[2])
                self.send_header("Content-Length", str(len(res[1])))
                self.end_headers()
                self.wfile.write(res[1])

            def do_GET(self):
                self._serve("GET")

            def do_POST(self):
                self._serve("POST")

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
