#define KPRE 1
#include "kernarg_lat.hip"
