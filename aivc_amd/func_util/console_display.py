"""Console helpers with the reference's column format (src/func_util/console_display.py)."""
MSG_TYPE_LEN = 10
FN_NAME_LEN = 30
VAR_NAME_LEN = 30
FLAG_QUIET = False


def print_log_msg(msg_type, fn_name, var_name, var):
    if FLAG_QUIET:
        return
    print(('[' + str(msg_type) + ']').ljust(MSG_TYPE_LEN) + (' | ' + str(fn_name)).ljust(FN_NAME_LEN)
          + (' | ' + str(var_name)).ljust(VAR_NAME_LEN) + ' | ' + str(var))


def print_dic_content(dic_to_print, dic_name=''):
    if FLAG_QUIET:
        return
    print(str(dic_name))
    for k, v in dic_to_print.items():
        if isinstance(v, dict):
            print_dic_content(v, dic_name=k)
        else:
            print('  %s: %s' % (k, v))
