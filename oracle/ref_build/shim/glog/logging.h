// NOT glog: open3d_slam/src/Transform.cpp includes this header and uses nothing of it
#pragma once
