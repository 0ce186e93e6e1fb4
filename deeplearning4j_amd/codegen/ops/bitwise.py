"""Bitwise namespace (reference definition: codegen/op-codegen/src/main/ops/org/nd4j/codegen/ops/Bitwise.kt)."""
from ..dsl import DataType as D, Namespace

ns = Namespace("bitwise", doc="Bit-level operations on integer arrays.", className="Bitwise")


def _binary(name, lib, doc, aliases=()):
    (ns.op(name, libnd4j=lib, doc=doc).input("x", D.INT, "First input array").input("y", D.INT, "Second input array")
       .output("output", D.INT, "Result, same shape as the (broadcast) inputs").alias(*aliases))


_binary("and", "bitwise_and", "Bitwise AND of two integer arrays (broadcastable).", ("bitsAnd",))
_binary("or", "bitwise_or", "Bitwise OR of two integer arrays (broadcastable).", ("bitsOr",))
_binary("xor", "bitwise_xor", "Bitwise XOR of two integer arrays (broadcastable).", ("bitsXor",))
_binary("bitsHammingDistance", "bits_hamming_distance", "Number of differing bits between the two arrays, summed over all elements.")
(ns.op("leftShift", libnd4j="shift_bits", doc="Shift the bits of x left by the per-element amounts in shift.")
   .input("x", D.INT, "Input to be bit shifted").input("shift", D.INT, "Amount to shift each element by").output("output", D.INT, "x << shift"))
(ns.op("rightShift", libnd4j="rshift_bits", doc="Shift the bits of x right by the per-element amounts in shift.")
   .input("x", D.INT, "Input to be bit shifted").input("shift", D.INT, "Amount to shift each element by").output("output", D.INT, "x >> shift"))
(ns.op("leftShiftCyclic", libnd4j="cyclic_shift_bits", doc="Rotate the bits of x left: bits shifted out on the left re-enter on the right.")
   .input("x", D.INT, "Input to be bit shifted").input("shift", D.INT, "Amount to rotate each element by").output("output", D.INT, "rotl(x, shift)"))
(ns.op("rightShiftCyclic", libnd4j="cyclic_rshift_bits", doc="Rotate the bits of x right: bits shifted out on the right re-enter on the left.")
   .input("x", D.INT, "Input to be bit shifted").input("shift", D.INT, "Amount to rotate each element by").output("output", D.INT, "rotr(x, shift)"))
(ns.op("toggleBits", libnd4j="toggle_bits", doc="Flip every bit of the input (bitwise NOT).")
   .input("x", D.INT, "Input array").output("output", D.INT, "~x"))
