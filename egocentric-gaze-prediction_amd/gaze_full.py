"""Mirror of the reference's ``gaze_full.py`` CLI (gaze_full.py:10-118): the same 37 flags with the same defaults,
then SP -> AT -> (extraction) -> LF.  Run as ``python -m egaze_amd.gaze_full --train_sp --train_lstm --train_late ...``
(one process per GPU; under ``torch.distributed.run`` the SP / LF optimizers all-reduce their gradients over RCCL).
"""
import argparse
import os

import torch
from torch.utils.data import DataLoader


def build_parser():
    p = argparse.ArgumentParser()
    a = p.add_argument
    a('--lr_late', type=float, default=1e-4, required=False, help='lr for LF Adam')
    a('--lr', type=float, default=1e-7, required=False, help='lr for SP Adam')
    a('--sp_resume', default='0', required=False, help='2 from fusion, 0 from vgg, 1 from separately trained models.')
    a('--sp_save_img', default='loss_SP.png', required=False)
    a('--late_save_img', default='loss_late.png', required=False)
    a('--pretrained_spatial', default='save/04_spatial.pth.tar', required=False)
    a('--pretrained_temporal', default='save/03_temporal.pth.tar', required=False)
    a('--pretrained_model', default=None, required=False, help='pretrained SP module')
    a('--pretrained_lstm', default=None, required=False, help='pretrained LSTM in AT module')
    a('--pretrained_late', default=None, required=False, help='pretrained LF module')
    a('--lstm_save_img', default='loss_lstm.png', required=False)
    a('--save_sp', default='best_SP.pth.tar', required=False)
    a('--save_lstm', default='best_lstm.pth.tar', required=False)
    a('--save_late', default='best_late.pth.tar', required=False)
    a('--save_path', default='save', required=False)
    a('--loss_function', default='f', required=False, help='if is not set as f, use bce loss')
    a('--num_epoch', type=int, default=10, required=False)
    a('--num_epoch_lstm', type=int, default=120, required=False)
    a('--extract_lstm', action='store_true')
    a('--extract_lstm_path', default='../512w', required=False)
    a('--train_sp', action='store_true')
    a('--train_lstm', action='store_true')
    a('--train_late', action='store_true')
    a('--extract_late', action='store_true')
    a('--extract_late_pred_folder', default='../new_pred/', required=False)
    a('--extract_late_feat_folder', default='../new_feat/', required=False)
    a('--device', default='0', help='GPU index of this process (one process per GPU)')
    a('--val_name', default='Alireza', required=False, help='cross subject validation')
    a('--task', default=None, required=False, help='cross task validation')
    a('--flowPath', default='../gtea_imgflow', required=False)
    a('--imagePath', default='../gtea_images', required=False)
    a('--fixsacPath', default='fixsac', required=False)
    a('--gtPath', default='../gtea_gts', required=False)
    a('--batch_size', type=int, default=64, help='batch size of LF')
    a('--batch_size_sp', type=int, default=8, help='batch size of SP')
    a('--crop_size', type=int, default=3, help='crop size of vgg conv5_3 feature')
    a('--align', action='store_true')
    return p


def _split(folder, val_name):
    names = os.listdir(folder)
    return sorted(k for k in names if val_name not in k), sorted(k for k in names if val_name in k)


def _at_stage(args, STTrainData, STValData):
    from .AT import AT
    att = AT(pretrained_model=args.pretrained_model, pretrained_lstm=args.pretrained_lstm,
             extract_lstm=args.extract_lstm, crop_size=args.crop_size, num_epoch_lstm=args.num_epoch_lstm,
             lstm_save_img=args.lstm_save_img, save_path=args.save_path, save_name=args.save_lstm,
             device=args.device, lstm_data_path=args.extract_lstm_path, traindata=STTrainData, valdata=STValData,
             task=args.task, align=args.align)
    if args.train_lstm:
        att.train()
    if args.extract_late:
        if not args.train_lstm:
            att.reload_LSTM(os.path.join(args.save_path, args.save_lstm))
        for data in (STValData, STTrainData):
            att.extract_late(DataLoader(dataset=data, batch_size=1, shuffle=False, num_workers=1, pin_memory=True),
                             args.extract_late_pred_folder, args.extract_late_feat_folder)


def _wait_for_rank0(key, poll_s=5.0):
    """Host-side rendezvous after a rank-0-only stage: rank 0 sets ``key`` in the default process group's store when it is
    done (or ``key + '/failed'`` on its way out of an exception), the others poll with sleep.  No device collective runs
    while waiting, and a failure of rank 0 ends the other ranks instead of leaving them in a barrier."""
    import time
    from . import dp
    if dp.world_size() == 1:
        return
    store = torch.distributed.distributed_c10d._get_default_store()
    if dp.is_main():
        store.set(key, "1")
        return
    while True:
        if store.check([key + "/failed"]):
            raise RuntimeError("rank 0 failed in its sequential stage (%s)" % key)
        if store.check([key]):
            return
        time.sleep(poll_s)


def main(argv=None):
    args = build_parser().parse_args(argv)
    from .AT import AT
    from .LF import LF
    from .SP import SP
    from .data.STdatas import STDataset
    from . import dp
    if 'LOCAL_RANK' in os.environ and int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import datetime
        args.device = os.environ['LOCAL_RANK']
        torch.cuda.set_device(int(args.device))
        # the AT stage below is sequential (rank 0 only) and takes hours: the other ranks wait for it on the HOST
        # (_wait_for_rank0: a key in the process group's store, polled with sleep) -- not inside a device collective, which
        # would spin for hours and need a multi-day collective timeout that also hides real hangs of the SP / LF all-reduces
        torch.distributed.init_process_group(os.environ.get('EGAZE_DIST_BACKEND', 'nccl'),
                                             timeout=datetime.timedelta(minutes=30))
    listFolders = sorted(os.listdir(args.flowPath))
    listGtFiles, listValGtFiles = _split(args.gtPath, args.val_name)
    print('num of training samples: ', len(listGtFiles))
    listfixsacTrain, listfixsacVal = _split(args.fixsacPath, args.val_name)
    listTrainFiles, listValFiles = _split(args.imagePath, args.val_name)
    print('num of val samples: ', len(listValFiles))
    STTrainData = STDataset(args.flowPath, args.imagePath, args.gtPath, listFolders, listTrainFiles, listGtFiles,
                            listfixsacTrain, args.fixsacPath, raw_u8=True)      # bytes over PCIe, normalised on the GPU
    STValData = STDataset(args.flowPath, args.imagePath, args.gtPath, listFolders, listValFiles, listValGtFiles,
                          listfixsacVal, args.fixsacPath, raw_u8=True)
    os.makedirs(args.save_path, exist_ok=True)
    if args.train_sp:
        sp = SP(lr=args.lr, loss_save=args.sp_save_img, save_name=args.save_sp, save_path=args.save_path,
                loss_function=args.loss_function, num_epoch=args.num_epoch, batch_size=args.batch_size_sp,
                device=args.device, resume=args.sp_resume, pretrained_spatial=args.pretrained_spatial,
                pretrained_temporal=args.pretrained_temporal, traindata=STTrainData, valdata=STValData)
        sp.train()
        args.pretrained_model = os.path.join(args.save_path, args.save_sp)
    # AT is a batch-1, sequence-1 recurrence whose hidden state is carried from sample to sample across the whole
    # dataset (AT.py:127-145, 199-253): it does not shard without changing its results, so under torch.distributed it
    # runs on rank 0 only (which also owns every file it writes) while the other ranks wait.
    if dp.is_main():
        try:
            _at_stage(args, STTrainData, STValData)
        except BaseException:
            if dp.world_size() > 1:       # let the waiting ranks go down with this one instead of polling for ever
                torch.distributed.distributed_c10d._get_default_store().set("at_stage_done/failed", "1")
            raise
    _wait_for_rank0("at_stage_done")
    lf = LF(pretrained_model=args.pretrained_late, save_path=args.save_path, late_save_img=args.late_save_img,
            save_name=args.save_late, device=args.device, late_pred_path=args.extract_late_pred_folder,
            num_epoch=args.num_epoch, late_feat_path=args.extract_late_feat_folder, gt_path=args.gtPath,
            val_name=args.val_name, batch_size=args.batch_size, loss_function=args.loss_function, lr=args.lr_late,
            task=args.task)
    if args.train_late:
        lf.train()
    else:
        lf.val()


if __name__ == '__main__':
    main()
