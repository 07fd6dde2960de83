// empty stand-in (syntax check only; NOT pcl_conversions)
#pragma once
