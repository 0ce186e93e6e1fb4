"""Recurrent namespace (reference definition: codegen/op-codegen/src/main/ops/org/nd4j/codegen/ops/RNN.kt). The configurable lstmLayer
(optional inputs selected by flags) keeps its hand-written front end in autodiff/namespaces.py; the fixed-signature cells are generated."""
from ..dsl import DataType as D, Namespace

ns = Namespace("rnn", doc="Recurrent cells and layers.", className="RNN")

(ns.op("gru", doc="GRU layer over a whole sequence.").input("x", D.NUMERIC, "Input [time, batch, nIn]").input("hLast", D.NUMERIC, "Initial state [batch, nOut]")
   .input("Wx", D.NUMERIC, "Input weights [nIn, 3 nOut]").input("Wh", D.NUMERIC, "Recurrent weights [nOut, 3 nOut]").input("biases", D.NUMERIC, "Biases [3 nOut]")
   .output("h", D.NUMERIC, "States for every step [time, batch, nOut]"))
(ns.op("gruCell", doc="One GRU step.").input("x", D.NUMERIC, "Input [batch, nIn]").input("hLast", D.NUMERIC, "Previous state").input("Wru", D.NUMERIC, "Reset / update weights")
   .input("Wc", D.NUMERIC, "Candidate weights").input("bru", D.NUMERIC, "Reset / update biases").input("bc", D.NUMERIC, "Candidate bias")
   .output("r", D.NUMERIC, "Reset gate").output("u", D.NUMERIC, "Update gate").output("c", D.NUMERIC, "Candidate").output("h", D.NUMERIC, "New state"))
(ns.op("lstmCell", libnd4j="lstmBlockCell", doc="One LSTM step (block form with optional peepholes).").input("x", D.NUMERIC, "Input [batch, nIn]")
   .input("cLast", D.NUMERIC, "Previous cell state").input("yLast", D.NUMERIC, "Previous output").input("W", D.NUMERIC, "Weights [(nIn + nOut), 4 nOut]")
   .input("Wci", D.NUMERIC, "Input-gate peephole").input("Wcf", D.NUMERIC, "Forget-gate peephole").input("Wco", D.NUMERIC, "Output-gate peephole").input("b", D.NUMERIC, "Biases")
   .arg("peephole", D.BOOL, "Use the peephole connections", default=False, slot="i", convert="int")
   .arg("forgetBias", D.FLOATING_POINT, "Added to the forget gate pre-activation", default=0.0).arg("clippingCellValue", D.FLOATING_POINT, "Cell clipping (0 = none)", default=0.0)
   .output("i", D.NUMERIC, "Input gate").output("c", D.NUMERIC, "Cell state").output("f", D.NUMERIC, "Forget gate").output("o", D.NUMERIC, "Output gate")
   .output("z", D.NUMERIC, "Block input").output("h", D.NUMERIC, "Cell state after tanh").output("y", D.NUMERIC, "Output"))
(ns.op("sru", doc="Simple recurrent unit over a sequence.").input("x", D.NUMERIC, "Input").input("initialC", D.NUMERIC, "Initial cell state... see nativeOrder")
   .input("weights", D.NUMERIC, "Weights").input("bias", D.NUMERIC, "Biases").nativeOrder("x", "weights", "bias", "initialC")
   .output("h", D.NUMERIC, "Outputs").output("c", D.NUMERIC, "Cell states"))
(ns.op("sruCell", doc="One SRU step.").input("x", D.NUMERIC, "Input").input("cLast", D.NUMERIC, "Previous cell state").input("weights", D.NUMERIC, "Weights")
   .input("bias", D.NUMERIC, "Biases").output("h", D.NUMERIC, "Output").output("c", D.NUMERIC, "Cell state"))
