// placeholder, replaced below
#include "hmt_internal.hpp"
