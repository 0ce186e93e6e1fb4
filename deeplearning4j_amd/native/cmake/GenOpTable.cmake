# Regenerates csrc/op_table.inc (declared contracts of every custom op) when the generator changes.
# Reference role: cmake/GenCompilation.cmake + the op-registration macros scraped by codegen/libnd4j-gen.
find_package(Python3 COMPONENTS Interpreter)
set(_gen ${CMAKE_CURRENT_SOURCE_DIR}/../../tools/gen_op_table.py)
if(Python3_FOUND AND EXISTS ${_gen})
  add_custom_command(
    OUTPUT ${CMAKE_CURRENT_SOURCE_DIR}/csrc/op_table.inc
    COMMAND ${Python3_EXECUTABLE} ${_gen}
    DEPENDS ${_gen}
    WORKING_DIRECTORY ${CMAKE_CURRENT_SOURCE_DIR}/../..
    COMMENT "Generating op_table.inc")
  add_custom_target(op_table DEPENDS ${CMAKE_CURRENT_SOURCE_DIR}/csrc/op_table.inc)
endif()
