"""Random namespace (reference definition: codegen/op-codegen/src/main/ops/org/nd4j/codegen/ops/Random.kt)."""
from ..dsl import DataType as D, Input, Namespace

ns = Namespace("random", doc="Random number generation; every op takes the output shape as trailing integers.", className="Random")


def _shape(op):
    i = Input("shape", D.LONG, "Shape of the new random array")
    i.kind = "shape"
    op.inputs.append(i)
    return op


_shape(ns.op("uniform", libnd4j="randomuniform", doc="Uniform distribution U(min, max).")
       .arg("min", D.FLOATING_POINT, "Lower bound").arg("max", D.FLOATING_POINT, "Upper bound").arg("datatype", D.DATA_TYPE, "Data type of the output", default="FLOAT")
       .output("output", D.NUMERIC, "Random array"))
_shape(ns.op("normal", libnd4j="random_normal", doc="Normal distribution N(mean, stddev^2).")
       .arg("mean", D.FLOATING_POINT, "Mean").arg("stddev", D.FLOATING_POINT, "Standard deviation").arg("datatype", D.DATA_TYPE, "Data type of the output", default="FLOAT")
       .output("output", D.NUMERIC, "Random array"))
_shape(ns.op("bernoulli", libnd4j="random_bernoulli", doc="Bernoulli distribution: 1 with probability p, else 0.")
       .arg("p", D.FLOATING_POINT, "Probability of 1").output("output", D.NUMERIC, "Random array"))
_shape(ns.op("exponential", libnd4j="random_exponential", doc="Exponential distribution with rate lambda.")
       .arg("lambda", D.FLOATING_POINT, "Rate").output("output", D.NUMERIC, "Random array"))
