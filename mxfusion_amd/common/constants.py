SET_PARAMETER_PREFIX = 'SET_'   # mxfusion/common/constants.py:16
