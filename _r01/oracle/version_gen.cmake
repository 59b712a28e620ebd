# Instantiates the reference's own include/SZ3/version.hpp.in with cmake's configure_file()
# (script mode: `cmake -DREF=... -DOUT=... -P version_gen.cmake`), using the values the
# reference's CMakeLists.txt sets at :2 (project(SZ3 VERSION 3.3.2)) and :7 (SZ3_DATA_VERSION 3.3.2).
# We do NOT run the reference's build system (it would FetchContent zstd from the network).
set(PROJECT_NAME SZ3)
set(PROJECT_VERSION 3.3.2)
set(PROJECT_VERSION_MAJOR 3)
set(PROJECT_VERSION_MINOR 3)
set(PROJECT_VERSION_PATCH 2)
set(PROJECT_VERSION_TWEAK 0)
set(SZ3_DATA_VERSION 3.3.2)
configure_file(${REF}/include/SZ3/version.hpp.in ${OUT}/SZ3/version.hpp)
