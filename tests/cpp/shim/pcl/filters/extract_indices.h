// empty stand-in: point_cloud_reflector_detect.h includes it, its declarations use nothing from it (syntax check only; NOT PCL)
#pragma once
