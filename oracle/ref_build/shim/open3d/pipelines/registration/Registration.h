// NOT Open3D: the namespace helpers.cpp aliases; nothing of it is used by the functions this build runs
#pragma once
namespace open3d {
namespace pipelines {
namespace registration {
class RegistrationResult;
}
}
}
