checkpoint_config = dict(interval=1)
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook')])
custom_hooks = []
dist_params = dict(backend='nccl')   # RCCL on ROCm
log_level = 'INFO'
load_from = None
resume_from = None
workflow = [('train', 1)]
