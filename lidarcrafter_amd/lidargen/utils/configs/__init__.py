from ._schema import (KITTI_Config_, MeanFlow_NUSC_Config, NUSC_Auto_Reg_Config,
                      NUSC_Auto_Reg_V2_Config, NUSC_Box_Layout_Config, NUSC_Box_Layout_V1_Config,
                      NUSC_Box_Layout_V2_Config, NUSC_Box_Layout_V3_Config,
                      NUSC_Box_Layout_V4_Config, NUSC_Box_Layout_V5_Config,
                      NUSC_Box_Layout_V6_Config, NUSC_Config, NUSC_HDIT_Config,
                      NUSC_Layout_Config, NUSC_Object_Config)

# same 15 names as the reference registry (lidargen/utils/configs/__init__.py:17-32)
__all__ = {
    "kitti-360": KITTI_Config_,
    "nuscenes-unet-uncond": NUSC_Config,
    "nuscenes-hdit-uncond": NUSC_HDIT_Config,
    "nuscenes-auto-reg": NUSC_Auto_Reg_Config,
    "nuscenes-auto-reg-v2": NUSC_Auto_Reg_V2_Config,
    "nuscenes-box-layout": NUSC_Box_Layout_Config,
    "nuscenes-box-layout-v1": NUSC_Box_Layout_V1_Config,
    "nuscenes-box-layout-v2": NUSC_Box_Layout_V2_Config,
    "nuscenes-box-layout-v3": NUSC_Box_Layout_V3_Config,
    "nuscenes-box-layout-v4": NUSC_Box_Layout_V4_Config,
    "nuscenes-box-layout-v5": NUSC_Box_Layout_V5_Config,
    "nuscenes-box-layout-v6": NUSC_Box_Layout_V6_Config,
    "meanflow-nusc": MeanFlow_NUSC_Config,
    "nuscenes-layout": NUSC_Layout_Config,
    "nuscenes-object": NUSC_Object_Config,
}
