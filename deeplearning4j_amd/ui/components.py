"""UI components (reference: deeplearning4j-ui-parent/deeplearning4j-ui-components/src/main/java/org/deeplearning4j/ui/
{api/{Component,Style,LengthUnit}.java, components/chart/{Chart,ChartLine,ChartScatter,ChartHistogram,ChartHorizontalBar,ChartStackedArea,
ChartTimeline}.java + style/StyleChart.java, components/table/ComponentTable.java, components/text/ComponentText.java,
components/component/ComponentDiv.java, components/decorator/DecoratorAccordion.java, standalone/StaticPageUtil.java}).

Components are plain data (builder pattern, JSON round trip with the reference's wrapper-object convention ``{"ChartLine": {...}}``) and
``StaticPageUtil.renderHTML`` turns a list of them into ONE self-contained HTML page: the reference ships d3 + a TypeScript runtime as
assets; here a ~100-line inline renderer draws the same component types as SVG, so a report opens from disk with no server and no assets.
"""
from __future__ import annotations

import html
import json
from typing import Dict, List, Optional, Sequence


class LengthUnit:
    Px, Percent, CM, MM, In = "px", "%", "cm", "mm", "in"


class Style:
    """Base style: size, margins, background (Style.java)."""
    _fields = ("width", "height", "widthUnit", "heightUnit", "marginTop", "marginBottom", "marginLeft", "marginRight", "backgroundColor")

    def __init__(self, **kw):
        for f in self._fields:
            setattr(self, f, kw.pop(f, None))
        if kw:
            raise TypeError("%s: unknown style options %s" % (type(self).__name__, sorted(kw)))

    def toDict(self):
        return {f: getattr(self, f) for f in self._fields if getattr(self, f) is not None}

    @classmethod
    def fromDict(cls, d):
        return cls(**d) if d else None

    def css(self):
        out = []
        if self.width is not None:
            out.append("width:%s%s" % (self.width, self.widthUnit or LengthUnit.Px))
        if self.height is not None:
            out.append("height:%s%s" % (self.height, self.heightUnit or LengthUnit.Px))
        for side in ("Top", "Bottom", "Left", "Right"):
            v = getattr(self, "margin" + side)
            if v is not None:
                out.append("margin-%s:%spx" % (side.lower(), v))
        if self.backgroundColor:
            out.append("background:%s" % self.backgroundColor)
        return ";".join(out)


class StyleChart(Style):
    _fields = Style._fields + ("strokeWidth", "pointSize", "seriesColors", "axisStrokeWidth", "titleStyle")


class StyleTable(Style):
    _fields = Style._fields + ("columnWidths", "columnWidthUnit", "borderWidthPx", "headerColor", "whitespaceMode")


class StyleText(Style):
    _fields = Style._fields + ("font", "fontSize", "underline", "color", "whitespacePre")


class StyleDiv(Style):
    _fields = Style._fields + ("floatValue",)


class StyleAccordion(Style):
    _fields = Style._fields


_REGISTRY: Dict[str, type] = {}


class Component:
    """Base of every renderable element (Component.java); subclasses list their payload fields in ``_fields``."""
    _fields: Sequence[str] = ()
    _style_cls = Style

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        _REGISTRY[cls.__name__] = cls

    def __init__(self, style: Optional[Style] = None, **kw):
        self.style = style
        for f in self._fields:
            setattr(self, f, kw.pop(f, None))
        if kw:
            raise TypeError("%s: unknown fields %s" % (type(self).__name__, sorted(kw)))

    def getStyle(self):
        return self.style

    def toDict(self):
        body = {"componentType": type(self).__name__}
        for f in self._fields:
            v = getattr(self, f)
            if isinstance(v, Component):
                v = v.toDict()
            elif isinstance(v, (list, tuple)) and v and isinstance(v[0], Component):
                v = [c.toDict() for c in v]
            if v is not None:
                body[f] = v
        if self.style is not None:
            body["style"] = {type(self.style).__name__: self.style.toDict()}
        return {type(self).__name__: body}

    def toJson(self):
        return json.dumps(self.toDict())

    @staticmethod
    def fromDict(d):
        (name, body), = d.items()
        cls = _REGISTRY[name]
        kw = {}
        for f in cls._fields:
            v = body.get(f)
            if isinstance(v, dict) and len(v) == 1 and next(iter(v)) in _REGISTRY:
                v = Component.fromDict(v)
            elif isinstance(v, list) and v and isinstance(v[0], dict) and len(v[0]) == 1 and next(iter(v[0])) in _REGISTRY:
                v = [Component.fromDict(c) for c in v]
            kw[f] = v
        st = body.get("style")
        style = None
        if st:
            (sname, sbody), = st.items()
            style = {c.__name__: c for c in (Style, StyleChart, StyleTable, StyleText, StyleDiv, StyleAccordion)}[sname].fromDict(sbody)
        return cls(style=style, **kw)

    @staticmethod
    def fromJson(s):
        return Component.fromDict(json.loads(s))

    def __eq__(self, o):
        return type(o) is type(self) and o.toDict() == self.toDict()


class _Builder:
    def __init__(self, cls, **kw):
        self._cls, self._kw = cls, kw

    def __getattr__(self, name):
        def setter(v):
            self._kw[name] = v
            return self
        return setter

    def build(self):
        return self._cls(**self._kw)


class Chart(Component):
    """Common chart options (Chart.java): title, axis limits, grid / legend switches."""
    _common = ("title", "suppressAxisHorizontal", "suppressAxisVertical", "showLegend", "setXMin", "setXMax", "setYMin", "setYMax", "gridVerticalStrokeWidth",
               "gridHorizontalStrokeWidth")
    _style_cls = StyleChart


class ChartLine(Chart):
    """Several named (x, y) series drawn as polylines."""
    _fields = Chart._common + ("seriesNames", "x", "y")

    class Builder(_Builder):
        def __init__(self, title=None, style=None):
            super().__init__(ChartLine, title=title, style=style, seriesNames=[], x=[], y=[])

        def addSeries(self, name, x, y):
            if len(x) != len(y):
                raise ValueError("ChartLine.addSeries: x and y lengths differ")
            self._kw["seriesNames"].append(name)
            self._kw["x"].append([float(v) for v in x])
            self._kw["y"].append([float(v) for v in y])
            return self


class ChartScatter(Chart):
    _fields = Chart._common + ("seriesNames", "x", "y")

    class Builder(ChartLine.Builder):
        def __init__(self, title=None, style=None):
            _Builder.__init__(self, ChartScatter, title=title, style=style, seriesNames=[], x=[], y=[])


class ChartHistogram(Chart):
    """Bins given by lower / upper bounds and a height."""
    _fields = Chart._common + ("lowerBounds", "upperBounds", "yValues")

    class Builder(_Builder):
        def __init__(self, title=None, style=None):
            super().__init__(ChartHistogram, title=title, style=style, lowerBounds=[], upperBounds=[], yValues=[])

        def addBin(self, lower, upper, y):
            self._kw["lowerBounds"].append(float(lower))
            self._kw["upperBounds"].append(float(upper))
            self._kw["yValues"].append(float(y))
            return self


class ChartHorizontalBar(Chart):
    _fields = Chart._common + ("labels", "values", "xmin", "xmax")

    class Builder(_Builder):
        def __init__(self, title=None, style=None):
            super().__init__(ChartHorizontalBar, title=title, style=style, labels=[], values=[])

        def addValue(self, name, value):
            self._kw["labels"].append(name)
            self._kw["values"].append(float(value))
            return self


class ChartStackedArea(Chart):
    _fields = Chart._common + ("x", "y", "labels")

    class Builder(_Builder):
        def __init__(self, title=None, style=None):
            super().__init__(ChartStackedArea, title=title, style=style, x=None, y=[], labels=[])

        def setXValues(self, x):
            self._kw["x"] = [float(v) for v in x]
            return self

        def addSeries(self, name, y):
            if self._kw["x"] is None or len(y) != len(self._kw["x"]):
                raise ValueError("ChartStackedArea.addSeries: set the x values first; y must match their length")
            self._kw["labels"].append(name)
            self._kw["y"].append([float(v) for v in y])
            return self


class ChartTimeline(Chart):
    """Lanes of [start, end] entries with a label and colour (ChartTimeline.java TimelineEntry)."""
    _fields = Chart._common + ("laneNames", "laneData")

    class Builder(_Builder):
        def __init__(self, title=None, style=None):
            super().__init__(ChartTimeline, title=title, style=style, laneNames=[], laneData=[])

        def addLane(self, name, entries):
            self._kw["laneNames"].append(name)
            self._kw["laneData"].append([{"entryLabel": e[0], "startTimeMs": int(e[1]), "endTimeMs": int(e[2]), "color": (e[3] if len(e) > 3 else None)}
                                         for e in entries])
            return self


class ComponentTable(Component):
    _fields = ("title", "header", "content")
    _style_cls = StyleTable

    class Builder(_Builder):
        def __init__(self, style=None):
            super().__init__(ComponentTable, style=style)

        def header(self, *h):
            self._kw["header"] = [str(v) for v in h]
            return self

        def content(self, rows):
            self._kw["content"] = [[str(v) for v in r] for r in rows]
            return self


class ComponentText(Component):
    _fields = ("text",)
    _style_cls = StyleText

    class Builder(_Builder):
        def __init__(self, text, style=None):
            super().__init__(ComponentText, text=text, style=style)


class ComponentDiv(Component):
    _fields = ("components",)
    _style_cls = StyleDiv

    def __init__(self, style=None, components=None, **kw):
        super().__init__(style=style, components=list(components or []), **kw)


class DecoratorAccordion(Component):
    _fields = ("title", "defaultCollapsed", "innerComponents")
    _style_cls = StyleAccordion

    class Builder(_Builder):
        def __init__(self, style=None):
            super().__init__(DecoratorAccordion, style=style, innerComponents=[], defaultCollapsed=False)

        def addComponents(self, *cs):
            self._kw["innerComponents"].extend(cs)
            return self


_RENDER_JS = r"""
const COLS=['#1f77b4','#d62728','#2ca02c','#9467bd','#ff7f0e','#8c564b','#e377c2','#7f7f7f'];
function el(t,a,txt){const e=document.createElementNS('http://www.w3.org/2000/svg',t);for(const k in a)e.setAttribute(k,a[k]);if(txt!==undefined)e.textContent=txt;return e;}
function frame(c,W,H){const s=el('svg',{viewBox:'0 0 '+W+' '+H,width:'100%'});if(c.title){s.appendChild(el('text',{x:W/2,y:14,'text-anchor':'middle','font-size':13,'font-weight':'bold'},c.title));}return s;}
function scale(lo,hi,a,b){const d=(hi-lo)||1;return v=>a+(b-a)*(v-lo)/d;}
function axes(s,x0,x1,y0,y1,W,H,p){s.appendChild(el('path',{d:'M'+p+' '+(H-p)+'H'+(W-p)+'M'+p+' '+(H-p)+'V'+p,stroke:'#444',fill:'none'}));
 [[p,H-p+12,x0],[W-p-20,H-p+12,x1],[2,H-p,y0],[2,p+4,y1]].forEach(a=>s.appendChild(el('text',{x:a[0],y:a[1],'font-size':10},(+a[2]).toPrecision(4))));}
function lim(c,xs,ys){return [c.setXMin??Math.min(...xs),c.setXMax??Math.max(...xs),c.setYMin??Math.min(...ys),c.setYMax??Math.max(...ys)];}
const R={
 ChartLine(c,scatter){const W=640,H=260,p=40,s=frame(c,W,H);const xs=(c.x||[]).flat(),ys=(c.y||[]).flat();if(!xs.length)return s;
  const [x0,x1,y0,y1]=lim(c,xs,ys),sx=scale(x0,x1,p,W-p),sy=scale(y0,y1,H-p,p);axes(s,x0,x1,y0,y1,W,H,p);
  (c.x||[]).forEach((xv,i)=>{if(scatter){xv.forEach((v,j)=>s.appendChild(el('circle',{cx:sx(v),cy:sy(c.y[i][j]),r:2.5,fill:COLS[i%8]})));}
   else s.appendChild(el('path',{d:xv.map((v,j)=>(j?'L':'M')+sx(v)+' '+sy(c.y[i][j])).join(''),stroke:COLS[i%8],fill:'none','stroke-width':1.5}));
   if(c.showLegend!==false)s.appendChild(el('text',{x:W-p-110,y:p+12*i,'font-size':10,fill:COLS[i%8]},(c.seriesNames||[])[i]||''));});return s;},
 ChartScatter(c){return R.ChartLine(c,true);},
 ChartHistogram(c){const W=640,H=260,p=40,s=frame(c,W,H);const lo=c.lowerBounds||[],hi=c.upperBounds||[],y=c.yValues||[];if(!lo.length)return s;
  const x0=Math.min(...lo),x1=Math.max(...hi),y1=Math.max(...y,1e-30),sx=scale(x0,x1,p,W-p),sy=scale(0,y1,H-p,p);axes(s,x0,x1,0,y1,W,H,p);
  lo.forEach((l,i)=>s.appendChild(el('rect',{x:sx(l),y:sy(y[i]),width:Math.max(sx(hi[i])-sx(l)-1,1),height:H-p-sy(y[i]),fill:COLS[0]})));return s;},
 ChartHorizontalBar(c){const n=(c.labels||[]).length,W=640,H=40+22*n,s=frame(c,W,H);const v=c.values||[];const x1=c.xmax??Math.max(...v,1e-30),x0=c.xmin??Math.min(0,...v),sx=scale(x0,x1,150,W-30);
  (c.labels||[]).forEach((l,i)=>{s.appendChild(el('text',{x:4,y:38+22*i,'font-size':11},l));s.appendChild(el('rect',{x:sx(Math.min(0,v[i])),y:26+22*i,width:Math.abs(sx(v[i])-sx(0)),height:16,fill:COLS[i%8]}));
   s.appendChild(el('text',{x:sx(Math.max(v[i],0))+4,y:38+22*i,'font-size':10},(+v[i]).toPrecision(4)));});return s;},
 ChartStackedArea(c){const W=640,H=260,p=40,s=frame(c,W,H);const x=c.x||[];if(!x.length)return s;const n=x.length,tot=new Array(n).fill(0);(c.y||[]).forEach(r=>r.forEach((v,j)=>tot[j]+=v));
  const x0=Math.min(...x),x1=Math.max(...x),y1=Math.max(...tot,1e-30),sx=scale(x0,x1,p,W-p),sy=scale(0,y1,H-p,p);axes(s,x0,x1,0,y1,W,H,p);const base=new Array(n).fill(0);
  (c.y||[]).forEach((r,i)=>{const top=r.map((v,j)=>base[j]+v);let d=top.map((v,j)=>(j?'L':'M')+sx(x[j])+' '+sy(v)).join('');for(let j=n-1;j>=0;j--)d+='L'+sx(x[j])+' '+sy(base[j]);
   s.appendChild(el('path',{d:d+'Z',fill:COLS[i%8],'fill-opacity':0.7}));s.appendChild(el('text',{x:W-p-110,y:p+12*i,'font-size':10,fill:COLS[i%8]},(c.labels||[])[i]||''));top.forEach((v,j)=>base[j]=v);});return s;},
 ChartTimeline(c){const lanes=c.laneData||[],W=640,H=40+26*lanes.length,s=frame(c,W,H);const all=lanes.flat();if(!all.length)return s;
  const t0=Math.min(...all.map(e=>e.startTimeMs)),t1=Math.max(...all.map(e=>e.endTimeMs)),sx=scale(t0,t1,120,W-10);
  lanes.forEach((ln,i)=>{s.appendChild(el('text',{x:4,y:40+26*i,'font-size':11},(c.laneNames||[])[i]||''));ln.forEach((e,j)=>{const r=el('rect',{x:sx(e.startTimeMs),y:26+26*i,width:Math.max(sx(e.endTimeMs)-sx(e.startTimeMs),1),height:18,fill:e.color||COLS[j%8]});
   r.appendChild(el('title',{},e.entryLabel+' ('+(e.endTimeMs-e.startTimeMs)+' ms)'));s.appendChild(r);});});return s;},
 ComponentTable(c){const t=document.createElement('table');if(c.header){const tr=t.insertRow();c.header.forEach(h=>{const th=document.createElement('th');th.textContent=h;tr.appendChild(th);});}
  (c.content||[]).forEach(r=>{const tr=t.insertRow();r.forEach(v=>tr.insertCell().textContent=v);});return t;},
 ComponentText(c){const d=document.createElement('div');d.textContent=c.text||'';return d;},
 ComponentDiv(c){const d=document.createElement('div');(c.components||[]).forEach(x=>d.appendChild(render(x)));return d;},
 DecoratorAccordion(c){const d=document.createElement('details');if(!c.defaultCollapsed)d.open=true;const s=document.createElement('summary');s.textContent=c.title||'';d.appendChild(s);
  (c.innerComponents||[]).forEach(x=>d.appendChild(render(x)));return d;}};
function css(st){if(!st)return'';const o=Object.values(st)[0]||{};let s='';if(o.width!=null)s+='width:'+o.width+(o.widthUnit||'px')+';';if(o.height!=null)s+='height:'+o.height+(o.heightUnit||'px')+';';
 if(o.backgroundColor)s+='background:'+o.backgroundColor+';';if(o.fontSize)s+='font-size:'+o.fontSize+'px;';if(o.color)s+='color:'+o.color+';';return s;}
function render(w){const name=Object.keys(w)[0],c=w[name];const e=(R[name]||R.ComponentText)(c);const st=css(c.style);if(st&&e.setAttribute)e.setAttribute('style',st);return e;}
"""


class StaticPageUtil:
    """Render components into one standalone HTML document (StaticPageUtil.java renderHTML / saveHTMLFile)."""

    @staticmethod
    def renderHTML(*components, title="DL4J-AMD report") -> str:
        comps: List[Component] = []
        for c in components:
            comps.extend(c if isinstance(c, (list, tuple)) else [c])
        data = json.dumps([c.toDict() for c in comps]).replace("</", "<\\/")
        return ("<!doctype html><html><head><meta charset=\"utf-8\"><title>%s</title><style>body{font-family:sans-serif;margin:20px}"
                "table{border-collapse:collapse;font-size:13px;margin:8px 0}td,th{border:1px solid #ccc;padding:3px 10px;text-align:left}"
                "summary{font-weight:bold;cursor:pointer;margin:6px 0}</style></head><body><div id=\"root\"></div><script>%s\n"
                "const DATA=%s;const root=document.getElementById('root');DATA.forEach(c=>root.appendChild(render(c)));</script></body></html>"
                % (html.escape(title), _RENDER_JS, data))

    @staticmethod
    def saveHTMLFile(path, *components, title="DL4J-AMD report"):
        with open(path, "w", encoding="utf-8") as f:
            f.write(StaticPageUtil.renderHTML(*components, title=title))
