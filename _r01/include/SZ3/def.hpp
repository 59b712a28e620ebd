// forwarding header: everything lives in SZ3/api/sz.hpp (C++ face of libsz3hip.so)
#include "api/sz.hpp"
