#include "../../reference_interface.h"   // test scaffolding: see that file
