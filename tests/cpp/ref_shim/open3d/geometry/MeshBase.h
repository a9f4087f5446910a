#include "PointCloud.h"
