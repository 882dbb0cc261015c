# hexl-fpgaConfig.cmake -- package config for the MI355X build of the hexl-fpga API.
# Plays the role of the reference's cmake/hexl-fpga/hexl-fpgaConfig.cmake (used by examples/ and
# experimental/bridge-seal through find_package(hexl-fpga)): defines the imported target `hexl-fpga::hexl-fpga`
# (alias `hexl-fpga`) pointing at libhexl-fpga.so and the public header directory of this repository.
#   cmake -Dhexl-fpga_DIR=<repo>/cmake/hexl-fpga ...
get_filename_component(_HEXL_MI355X_ROOT "${CMAKE_CURRENT_LIST_DIR}/../.." ABSOLUTE)
set(_HEXL_MI355X_LIB "${_HEXL_MI355X_ROOT}/hexl-fpga_amd/lib/libhexl-fpga.so")
if(NOT EXISTS "${_HEXL_MI355X_LIB}")
  message(FATAL_ERROR "libhexl-fpga.so not built: run `make -C ${_HEXL_MI355X_ROOT}/hexl-fpga_amd/csrc && "
                      "make -C ${_HEXL_MI355X_ROOT}/hexl-fpga_amd/host`")
endif()
if(NOT TARGET hexl-fpga::hexl-fpga)
  add_library(hexl-fpga::hexl-fpga SHARED IMPORTED)
  set_target_properties(hexl-fpga::hexl-fpga PROPERTIES
    IMPORTED_LOCATION "${_HEXL_MI355X_LIB}"
    INTERFACE_INCLUDE_DIRECTORIES "${_HEXL_MI355X_ROOT}/include")
  add_library(hexl-fpga ALIAS hexl-fpga::hexl-fpga)
endif()
set(hexl-fpga_FOUND TRUE)
set(hexl-fpga_INCLUDE_DIRS "${_HEXL_MI355X_ROOT}/include")
set(hexl-fpga_LIBRARIES hexl-fpga::hexl-fpga)
